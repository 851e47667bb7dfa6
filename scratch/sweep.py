"""timing sweep: python scratch/sweep.py  (env FEWBIT_HIP_LIB / FEWBIT_HIP_WAVES_PER_CU select the variant)"""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from fewbit_amd import cabi
from tests.helpers import from_raw
z = np.load('tests/golden/quantize_ref.npz')
dev = 'cuda'
def table(name, k, dtype):
    tag = {torch.float32: 'f32', torch.bfloat16: 'bf16', torch.float16: 'f16'}[dtype]
    return from_raw(z[f'{name}{k:02d}_{tag}_borders'], dtype).to(dev), from_raw(z[f'{name}{k:02d}_{tag}_levels'], dtype).to(dev)
def timeit(fn, iters=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters
cases = [(torch.bfloat16, 3, 4096 * 4096, 'gelu')]
if len(sys.argv) > 1 and sys.argv[1] == 'all':
    cases += [(torch.float16, 2, 8192 * 8192, 'gelu'), (torch.float16, 4, 8192 * 8192, 'gelu'), (torch.float32, 3, 4096 * 4096, 'gelu'), (torch.bfloat16, 3, 16384 * 4096 // 2, 'gelu')]
tag = f"lib={os.path.basename(os.environ.get('FEWBIT_HIP_LIB', 'default'))} wpc={os.environ.get('FEWBIT_HIP_WAVES_PER_CU', '32')}"
for dtype, k, n, fnname in cases:
    b, l = table('gelu', k, dtype)
    x = torch.randn(n, device=dev).to(dtype); gy = torch.randn(n, device=dev).to(dtype)
    y = torch.empty_like(x); gx = torch.empty_like(x)
    st = torch.empty(cabi.state_nbytes(n, k), dtype=torch.uint8, device=dev)
    es = x.element_size()
    tf = timeit(lambda: cabi.quantize_forward(fnname, x, b, out=y, state=st))
    tb = timeit(lambda: cabi.quantize_backward(gy, st, l, out=gx))
    tc = timeit(lambda: y.copy_(x))
    tot = n * (4 * es + k / 4)
    print(f'{tag} {str(dtype)[6:]} k={k} n={n}: fwd {tf:.2f} us ({n*(2*es+k/8)/tf/1e6:.2f} TB/s)  bwd {tb:.2f} us ({n*(2*es+k/8)/tb/1e6:.2f} TB/s)  '
          f'fwd+bwd {tf+tb:.2f} us = {tot/(tf+tb)/1e6:.2f} TB/s = {tot/(tf+tb)/8e6*100:.1f}% of 8TB/s   [copy {tc:.2f} us {2*n*es/tc/1e6:.2f} TB/s]')
