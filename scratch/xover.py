"""search kernel vs pattern-table kernel by tensor size (16-bit dtypes): run twice, FEWBIT_HIP_LUT_MIN=1 (table always)
and FEWBIT_HIP_LUT_MIN=999999999999 (never); the threshold in lut_min_elements() is where the two cross"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store
dev='cuda'
def timeit(f, iters=2000):
    for _ in range(2500): f()          # settle (clock transient after idle)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1)*1000/iters)
    return best
tag = 'table' if int(os.environ.get('FEWBIT_HIP_LUT_MIN', '0')) == 1 else 'search'
for name, k, dtype in (('gelu', 3, torch.bfloat16), ('silu', 4, torch.float16)):
    b,_ = store.get(name, k, dev, dtype); b=b[1:-1].contiguous()
    for n in (1<<20, 1<<21, 3<<20, 1<<22, 5<<20, 3<<21, 7<<20, 1<<23):
        x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x); st = torch.empty(cabi.state_nbytes(n,k), dtype=torch.uint8, device=dev)
        f = cabi.bind_forward(name, x, b, out=y, state=st)
        print(f'{tag:6s} {name}{k} {str(dtype)[6:]:8s} n={n/2**20:4.1f}Mi: {timeit(f):6.2f} us', flush=True)
