"""1-bit family: forward/backward time vs torch's relu / threshold_backward"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
dev='cuda'
def timeit(f, iters=500):
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1000/iters
for dtype, n in ((torch.float32, 1024*1024), (torch.float32, 4096*4096), (torch.bfloat16, 4096*4096), (torch.float16, 8192*8192)):
    es = torch.empty(0, dtype=dtype).element_size()
    x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x); gy = torch.randn(n, device=dev).to(dtype); gx = torch.empty_like(x)
    for name in ('relu', 'leaky_relu', 'hardtanh'):
        p = {'leaky_relu': (0.01, 0.0), 'hardtanh': (-1.0, 1.0)}.get(name, (0.0, 0.0))
        st = torch.empty((n + 7) // 8, dtype=torch.uint8, device=dev)
        f = lambda: cabi.stepwise1_forward(name, x, *p, out=y, state=st)
        b = lambda: cabi.stepwise1_backward(name, gy, st, p[0], out=gx)
        tf, tb = timeit(f), timeit(b)
        byts = n * (2 * es + 1 / 8)
        print(f'{str(dtype)[6:]} n={n} {name}: fwd {tf:.1f} us ({byts/tf/1e6:.2f} TB/s) bwd {tb:.1f} us ({byts/tb/1e6:.2f} TB/s)', flush=True)
    tr = timeit(lambda: torch.relu(x)); 
    yy = torch.relu(x)
    tb = timeit(lambda: torch.ops.aten.threshold_backward(gy, yy, 0.0))
    print(f'   torch relu fwd {tr:.1f} us ({n*2*es/tr/1e6:.2f} TB/s) threshold_backward {tb:.1f} us ({n*3*es/tb/1e6:.2f} TB/s)')
