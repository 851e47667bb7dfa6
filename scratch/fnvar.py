"""forward time by functor at 4096x4096 bf16, 3-bit table (pattern-table kernel): how much of the forward is activation math"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store
dev='cuda'
def timeit(f, iters=1500):
    for _ in range(50): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1000/iters
n = 4096*4096
for dtype in (torch.bfloat16, torch.float16):
    b, l = store.get('gelu', 3, dev, dtype); b = b[1:-1].contiguous()
    x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x); st = torch.empty(cabi.state_nbytes(n, 3), dtype=torch.uint8, device=dev)
    res = []
    for fn in ('identity', 'gelu', 'silu', 'sigmoid', 'tanh', 'softsign', 'hardswish', 'elu', 'mish', 'softplus', 'logsigmoid', 'tanhshrink'):
        p = {'elu': (1.0, 0.0), 'softplus': (1.0, 20.0)}.get(fn, (0.0, 0.0))
        f = cabi.bind_forward(fn, x, b, out=y, state=st, p0=p[0], p1=p[1])
        res.append(f'{fn} {timeit(f):.2f}')
    print(str(dtype)[6:], ' | '.join(res), flush=True)
