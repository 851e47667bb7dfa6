"""relu 1-bit fp32 forward / backward / step by size"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
dev = 'cuda'
def timeit(fns, rounds=400):
    for _ in range(300):
        for f in fns: f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(rounds):
            for f in fns: f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / rounds)
    return best
tag = os.path.basename(os.environ.get('FEWBIT_HIP_LIB', 'default'))
for dtype in (torch.float32, torch.bfloat16):
    for n in (1 << 20, 1 << 22, 1 << 24, 50331648):
        x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x); gy = torch.randn(n, device=dev).to(dtype); gx = torch.empty_like(x)
        st = torch.empty(cabi.state_nbytes(n, 1), dtype=torch.uint8, device=dev)
        f = cabi.bind_stepwise1_forward('relu', x, out=y, state=st); b = cabi.bind_stepwise1_backward('relu', gy, st, out=gx)
        es = x.element_size(); byts = n * (2 * es + 1 / 8)
        tf, tb, ts = timeit([f]), timeit([b]), timeit([f, b])
        print(f'{tag} relu {str(dtype)[6:]:8s} n={n:9d}: fwd {tf:7.2f} us ({byts/tf/8e4:5.1f}%) bwd {tb:7.2f} us ({byts/tb/8e4:5.1f}%) step {ts:7.2f} us ({2*byts/ts/8e4:5.1f}%)', flush=True)
