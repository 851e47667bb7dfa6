"""768-wide shape: 4-wave tile (128 x 256) with few slices against the planner's choice (8 waves, 6 slices)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewbit_amd import cabi

def timed(f, reps=40):
    for _ in range(10):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for rnd in range(2):
  for dist in ('rademacher', 'gaussian'):
    for rows, features, proj in ((16384, 768, 3276), (16384, 768, 1638), (16384, 3072, 3276)):
        for w, z in ((-1, -1), (4, 2), (4, 3), (4, 4), (4, 6), (8, 3), (8, 4), (8, 6)):
            cabi.tune_sketch_waves(w); cabi.tune_sketch_slices(z); cabi.tune_sketch_halves(1 if w != -1 else -1)
            m = torch.randn(rows, features, device='cuda').to(torch.bfloat16)
            plan = cabi.describe_sketch(dist, rows, features, proj)
            ws = torch.empty(max(plan['workspace_bytes'], 1), dtype=torch.uint8, device='cuda')
            o = torch.empty(proj, features, dtype=torch.bfloat16, device='cuda')
            us = timed(lambda: cabi.sketch(dist, m, proj, 1, 1.0, out=o, workspace=ws))
            g = plan['grid']
            print(dist, rows, features, proj, 'waves', w, 'grid', g, 'wgs', g[0] * g[1] * g[2], f'{us:.1f} us', f'{2 * rows * features * proj / us / 1e6:.0f} TFLOP/s', flush=True)
cabi.tune_sketch_slices(-1); cabi.tune_sketch_waves(-1); cabi.tune_sketch_halves(-1)
