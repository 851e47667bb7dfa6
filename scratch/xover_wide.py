import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
dev='cuda'
def timeit(f, iters=500):
    for _ in range(30): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1000/iters
dtype=torch.bfloat16
for nlev in (12, 32, 64, 256):
    b = torch.linspace(-3, 3, nlev - 1).to(dtype).to(dev)
    k = cabi.bitwidth(nlev)
    row = []
    for e in (17, 18, 19, 20, 21, 22, 23):
        n = 1 << e
        x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x); st = torch.empty(cabi.state_nbytes(n,k), dtype=torch.uint8, device=dev)
        f = cabi.bind_forward('gelu', x, b, out=y, state=st)
        row.append(f'2^{e}: {timeit(f):.2f}')
    print(f'nlev={nlev} k={k} LUT_MIN={os.environ.get("FEWBIT_HIP_LUT_MIN")}:', ' | '.join(row), flush=True)
