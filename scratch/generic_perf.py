import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
dev='cuda'
def timeit(f, iters=200):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1000/iters
n = 4096*4096
for dtype in (torch.bfloat16, torch.float32):
    es = torch.empty(0, dtype=dtype).element_size()
    for nlev in (8, 16, 32, 64, 256):
        b = torch.linspace(-3, 3, nlev - 1).to(dtype).to(dev); l = torch.rand(nlev).to(dtype).to(dev)
        k = cabi.bitwidth(nlev)
        x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x); gy = torch.randn(n, device=dev).to(dtype); gx = torch.empty_like(x)
        st = torch.empty(cabi.state_nbytes(n, k), dtype=torch.uint8, device=dev)
        f = cabi.bind_forward('gelu', x, b, out=y, state=st); bw = cabi.bind_backward(gy, st, l, out=gx)
        tf, tb = timeit(f), timeit(bw)
        byts = n * (2 * es + k / 8)
        print(f'{str(dtype)[6:]} k={k}: fwd {tf:.1f} us ({byts/tf/1e6:.2f} TB/s) bwd {tb:.1f} us ({byts/tb/1e6:.2f} TB/s)', flush=True)
    # misaligned k=3
    b = torch.linspace(-3, 3, 7).to(dtype).to(dev); l = torch.rand(8).to(dtype).to(dev)
    xx = torch.randn(n + 8, device=dev).to(dtype)[1:n+1]; yy = torch.empty(n + 8, dtype=dtype, device=dev)[1:n+1]
    st = torch.empty(cabi.state_nbytes(n, 3), dtype=torch.uint8, device=dev)
    f = cabi.bind_forward('gelu', xx, b, out=yy, state=st); bw = cabi.bind_backward(xx, st, l, out=yy)
    print(f'{str(dtype)[6:]} k=3 misaligned: fwd {timeit(f):.1f} us bwd {timeit(bw):.1f} us')
