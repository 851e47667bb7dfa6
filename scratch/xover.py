import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store
dev='cuda'
def timeit(f, iters=1000):
    for _ in range(30): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1000/iters
dtype=torch.bfloat16
for k in (3, 4):
    b,_ = store.get('gelu', k, dev, dtype); b=b[1:-1].contiguous()
    for n in (1<<17, 1<<18, 1<<19, 1<<20, 1<<21, 1<<22, 1<<23):
        x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x); st = torch.empty(cabi.state_nbytes(n,k), dtype=torch.uint8, device=dev)
        f = cabi.bind_forward('gelu', x, b, out=y, state=st)
        print(f'k={k} n=2^{n.bit_length()-1}: {timeit(f):.2f} us', flush=True)
