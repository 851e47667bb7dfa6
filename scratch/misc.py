import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store
dev='cuda'
def timeit(f, iters=500):
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1000/iters
for dtype in (torch.bfloat16, torch.float32):
    for n in (1024*1024, 4096*4096):
        x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x); st = torch.empty(n//8, dtype=torch.uint8, device=dev)
        es = x.element_size()
        for fn in ('relu', 'leaky_relu', 'hardtanh'):
            tf = timeit(lambda: cabi.stepwise1_forward(fn, x, 0.1, 1.0, out=y, state=st))
            tb = timeit(lambda: cabi.stepwise1_backward(fn, x, st, 0.1, out=y))
            byts = n*(2*es+0.125)
            print(f'{fn} {str(dtype)[6:]} n={n}: fwd {tf:.2f} us ({byts/tf/1e6:.2f} TB/s) bwd {tb:.2f} us ({byts/tb/1e6:.2f} TB/s)')
# all continuous fns, bf16 k=3 4096^2
n = 4096*4096
for dtype in (torch.bfloat16,):
    x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x); st = torch.empty(cabi.state_nbytes(n,3), dtype=torch.uint8, device=dev)
    for fn in cabi.CONTINUOUS:
        name = 'tanh' if fn == 'identity' else fn
        b,_ = store.get(name, 3, dev, dtype); b=b[1:-1].contiguous()
        tf = timeit(lambda: cabi.quantize_forward(fn, x, b, 1.0, 20.0, out=y, state=st), 200)
        print(f'{fn} bf16 k=3 fwd {tf:.2f} us ({n*(4+0.375)/tf/1e6:.2f} TB/s)')
# generic path: 5 levels
b = torch.tensor([-1.0,-0.2,0.3,1.1], device=dev, dtype=torch.bfloat16); l = torch.randn(5, device=dev).to(torch.bfloat16)
tf = timeit(lambda: cabi.quantize_forward('gelu', x, b, out=y, state=st), 100); tb = timeit(lambda: cabi.quantize_backward(x, st, l, out=y), 100)
print(f'generic 5 levels bf16: fwd {tf:.1f} us bwd {tb:.1f} us')
