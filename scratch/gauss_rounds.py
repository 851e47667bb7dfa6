"""Gaussian sketch: time per shape (the library FEWBIT_HIP_LIB points at)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewbit_amd import cabi

def timed(f, reps=30):
    for _ in range(10):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for rnd in range(2):
    row = []
    for rows, features, proj in ((16384, 768, 3276), (16384, 3072, 3276), (16384, 768, 1638), (16384, 3072, 1638), (65536, 4096, 4096)):
        m = torch.randn(rows, features, device='cuda').to(torch.bfloat16)
        ws = torch.empty(max(cabi.sketch_workspace_bytes('gaussian', rows, features, proj), 1), dtype=torch.uint8, device='cuda')
        o = torch.empty(proj, features, dtype=torch.bfloat16, device='cuda')
        us = timed(lambda: cabi.sketch('gaussian', m, proj, 1, 1.0, out=o, workspace=ws))
        row.append(f'{rows}x{features} p={proj}: {us:.1f} us = {2 * rows * features * proj / us / 1e6:.0f} TFLOP/s')
    print(os.environ.get('FEWBIT_HIP_LIB', 'production')[-12:], ' | '.join(row), flush=True)
