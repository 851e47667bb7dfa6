"""time one configuration with the library named by FEWBIT_HIP_LIB (ablation builds of fewbit_sketch.hip)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewbit_amd import cabi
dist, rows, features, proj = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
z = int(sys.argv[5]) if len(sys.argv) > 5 else -1
cabi.tune_sketch_slices(z)
w = int(sys.argv[6]) if len(sys.argv) > 6 else -1
cabi.tune_sketch_waves(w)
m = torch.randn(rows, features, device='cuda').to(torch.bfloat16)
plan = cabi.describe_sketch(dist, rows, features, proj)
ws = torch.empty(max(plan['workspace_bytes'], 1), dtype=torch.uint8, device='cuda')
o = torch.empty(proj, features, dtype=torch.bfloat16, device='cuda')
f = lambda: cabi.sketch(dist, m, proj, 1234, 1.0 / proj, out=o, workspace=ws)
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print('%-28s %s grid=%s thr=%d  %.1f us  %.0f TFLOP/s' % (os.path.basename(os.environ.get('FEWBIT_HIP_LIB', 'production')), dist, plan['grid'], plan['threads'], us, 2.0 * proj * rows * features / us / 1e6))
