"""one configuration of the random-projection kernel, a few launches: the program rocprofv3 wraps (tools/profile_sketch.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewbit_amd import cabi
dist = sys.argv[1] if len(sys.argv) > 1 else 'rademacher'
rows, features, proj = (int(a) for a in (sys.argv[2:5] if len(sys.argv) > 4 else (16384, 3072, 1638)))
dtype = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f16': torch.float16}[sys.argv[5] if len(sys.argv) > 5 else 'bf16']
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
if len(sys.argv) > 7:
    cabi.tune_sketch_slices(int(sys.argv[7]))
m = torch.randn(rows, features, device='cuda').to(dtype)
plan = cabi.describe_sketch(dist, rows, features, proj, dtype)
ws = torch.empty(max(plan['workspace_bytes'], 1), dtype=torch.uint8, device='cuda')
o = torch.empty(proj, features, dtype=dtype, device='cuda')
for _ in range(reps):
    cabi.sketch(dist, m, proj, 1234, 1.0 / proj, out=o, workspace=ws)
torch.cuda.synchronize()
print(plan)
