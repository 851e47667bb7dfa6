"""step / fwd / bwd time at the headline config (+ silu4 fp16 8192^2, gelu fp32) for the build in FEWBIT_HIP_LIB"""
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
tag = os.path.basename(os.environ.get('FEWBIT_HIP_LIB', 'default'))
out = []
for name, k, dtype, n in (('gelu', 3, torch.bfloat16, 4096*4096), ('silu', 4, torch.float16, 8192*8192), ('gelu', 3, torch.float32, 4096*4096)):
    b, l = store.get(name, k, dev, dtype); b = b[1:-1].contiguous()
    x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x); st = torch.empty(cabi.state_nbytes(n, k), dtype=torch.uint8, device=dev)
    gy = torch.randn(n, device=dev).to(dtype); gx = torch.empty_like(x)
    f = cabi.bind_forward(name, x, b, out=y, state=st); bw = cabi.bind_backward(gy, st, l, out=gx)
    def step(): f(); bw()
    step(); torch.cuda.synchronize()
    codes = cabi.unpack_codes(st, n, k).long()
    ok = torch.equal(codes, torch.bucketize(x.float(), b.float())) and torch.equal(gx.float(), (l[codes].float() * gy.float()).to(dtype).float())
    out.append(f'{name}{k}{str(dtype)[6:]} fwd {timeit(f):.2f} bwd {timeit(bw):.2f} step {timeit(step):.2f} ok={ok}')
print(tag, ' | '.join(out), flush=True)
