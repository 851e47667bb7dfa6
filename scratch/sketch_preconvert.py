"""fp32 input: the kernel's own staging conversion against one streaming conversion pass to bf16 followed by the bf16 kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewbit_amd import cabi

def timed(f, reps=30):
    for _ in range(8):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for rnd in range(2):
    for dist in ('rademacher', 'gaussian'):
        for rows, features, proj in ((16384, 768, 3276), (16384, 3072, 3276), (16384, 768, 1638), (16384, 3072, 1638)):
            m = torch.randn(rows, features, device='cuda')
            mb = torch.empty(rows, features, device='cuda', dtype=torch.bfloat16)
            t32 = timed(lambda: cabi.sketch(dist, m, proj, 1))
            tcv = timed(lambda: mb.copy_(m))
            t16 = timed(lambda: cabi.sketch(dist, mb, proj, 1))
            both = timed(lambda: cabi.sketch(dist, mb.copy_(m), proj, 1))
            print(f'{dist:10s} {rows}x{features} p={proj}: fp32 kernel {t32:6.1f} us | convert {tcv:5.1f} + bf16 kernel {t16:6.1f} = {tcv + t16:6.1f} (measured together {both:6.1f}) -> {100 * (1 - both / t32):.1f} % less', flush=True)
