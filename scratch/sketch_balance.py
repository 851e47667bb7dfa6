"""How much of the 768-wide sketch's gap is workgroup balance?  Shapes whose tiles x slices fill 256 CUs exactly beside the
RoBERTa shape (39 tiles x 6 slices = 234 workgroups, 21-22 K stages each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewbit_amd import cabi

def timed(f, reps=30):
    for _ in range(5):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for dist in ('rademacher', 'gaussian'):
    for rows, features, proj, z in ((16384, 768, 3276, -1), (16384, 768, 3328, -1), (15360, 768, 3328, 6), (16384, 1024, 2048, 8),
                                    (16384, 1024, 2048, 4), (16384, 2048, 2048, 4), (16384, 768, 3276, 3), (16384, 768, 3276, 12), (16384, 768, 3276, 13)):
        cabi.tune_sketch_slices(z)
        m = torch.randn(rows, features, device='cuda').to(torch.bfloat16)
        plan = cabi.describe_sketch(dist, rows, features, proj)
        ws = torch.empty(max(plan['workspace_bytes'], 1), dtype=torch.uint8, device='cuda')
        o = torch.empty(proj, features, dtype=torch.bfloat16, device='cuda')
        us = timed(lambda: cabi.sketch(dist, m, proj, 1, 1.0, out=o, workspace=ws))
        print(dist, rows, features, proj, 'grid', plan['grid'], 'wgs', plan['grid'][0] * plan['grid'][1] * plan['grid'][2], f'{us:.1f} us', f'{2 * rows * features * proj / us / 1e6:.0f} TFLOP/s', flush=True)
cabi.tune_sketch_slices(-1)
