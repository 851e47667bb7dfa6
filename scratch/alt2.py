import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from fewbit_amd import cabi
from fewbit_amd.store import store
dev = 'cuda'; n = 4096 * 4096; dt = torch.bfloat16
bo, lv = store.get('gelu', 3, dev, dt); bo = bo[1:-1].contiguous()
x = torch.randn(n, device=dev).to(dt); y = torch.empty_like(x); gy = torch.randn(n, device=dev).to(dt); gx = torch.empty_like(x)
st = torch.empty(cabi.state_nbytes(n, 3), dtype=torch.uint8, device=dev); st2 = torch.empty_like(st)
def timeit(fns, iters=2000):
    for _ in range(50):
        for f in fns: f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        for f in fns: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters
F = cabi.bind_forward('gelu', x, bo, out=y, state=st)
cabi.quantize_forward('gelu', x, bo, out=y, state=st2)
B = cabi.bind_backward(gy, st, lv, out=gx)
B2 = cabi.bind_backward(gy, st2, lv, out=gx)
Fi = cabi.bind_forward('gelu', x, bo, out=x, state=st)     # in place
print('F only            %.2f' % timeit([F]))
print('B only            %.2f' % timeit([B]))
print('F,B (dependent)   %.2f' % timeit([F, B]))
print('F,B2 (independent state) %.2f' % timeit([F, B2]))
print('F,F,B,B           %.2f per pair' % (timeit([F, F, B, B]) / 2))
print('F x4, B x4        %.2f per pair' % (timeit([F] * 4 + [B] * 4) / 4))
s2 = torch.cuda.Stream()
def two_streams():
    F()
    with torch.cuda.stream(s2):
        pass
