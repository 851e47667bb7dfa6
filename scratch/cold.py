"""cold (rotating >= 1 GiB of buffers) vs warm timings for several configs; env FEWBIT_HIP_LIB selects the build"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from fewbit_amd import cabi
from fewbit_amd.store import store
dev = 'cuda'
def timeit(fns, iters):
    for f in fns: f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        for f in fns: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters / len(fns)
tag = os.path.basename(os.environ.get('FEWBIT_HIP_LIB', 'default'))
for name, k, dtype, n in (('gelu', 3, torch.bfloat16, 4096 * 4096), ('silu', 2, torch.float16, 8192 * 8192), ('silu', 4, torch.float16, 8192 * 8192), ('gelu', 3, torch.float32, 4096 * 4096)):
    es = torch.empty(0, dtype=dtype).element_size()
    nsets = max(2, int(1.2 * 2**30 / (n * es * 2)) + 1)
    bo, lv = store.get(name, k, dev, dtype); bo = bo[1:-1].contiguous()
    sets = []
    for _ in range(nsets):
        x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x)
        st = torch.empty(cabi.state_nbytes(n, k), dtype=torch.uint8, device=dev); sets.append((x, y, st))
    F = [cabi.bind_forward(name, x, bo, out=y, state=st) for x, y, st in sets]
    B = [cabi.bind_backward(y, st, lv, out=x) for x, y, st in sets]
    it = max(20, 2000 // nsets)
    fw, bw = timeit([F[0]], 300), timeit([B[0]], 300)
    fc, bc = timeit(F, it), timeit(B, it)
    byts = n * (2 * es + k / 8)
    print(f'{tag} {name} k={k} {str(dtype)[6:]} n={n}: warm fwd {fw:.1f} bwd {bw:.1f} us ({2*byts/(fw+bw)/1e6:.2f} TB/s = {2*byts/(fw+bw)/8e4:.1f}%) | cold fwd {fc:.1f} bwd {bc:.1f} us ({2*byts/(fc+bc)/1e6:.2f} TB/s = {2*byts/(fc+bc)/8e4:.1f}%)')
    C = [(lambda a=x, b=y: b.copy_(a)) for x, y, st in sets]
    cw, cc = timeit([C[0]], 300), timeit(C, it)
    cb = n * 2 * es
    print(f'    torch copy_ of the same tensor: warm {cw:.1f} us ({cb/cw/1e6:.2f} TB/s) | cold {cc:.1f} us ({cb/cc/1e6:.2f} TB/s)')
    del sets, F, B, C
