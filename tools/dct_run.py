"""One configuration of the sampled cosine transform -- the program rocprofv3 wraps (tools/profile_dct.sh).

    python3 tools/dct_run.py <rows> <features> <proj> <bf16|f16|f32> [reps=200] [settle_ms=40] [seeded|explicit|torch]

Settles like tools/sketch_run.py (>= settle_ms of the same call first), then times `reps` calls between two HIP events.  Last
argument: `seeded` (default; fewbit_hip_sampled_dct_seeded, what the layer calls: the rows are a function of a seed), `explicit`
(fewbit_hip_sampled_dct on an int64 idx array), `torch` (the torch.fft formulation of the same result).  Prints one JSON line.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fewbit_amd import cabi, linear  # noqa: E402

rows, features, proj = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (16384, 768, 3276)))
dname = sys.argv[4] if len(sys.argv) > 4 else 'bf16'
dtype = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f16': torch.float16}[dname]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 200
settle_ms = float(sys.argv[6]) if len(sys.argv) > 6 else 40.0
mode = sys.argv[7] if len(sys.argv) > 7 else 'seeded'
assert mode in ('seeded', 'explicit', 'torch'), mode
use_torch = mode == 'torch'

m = torch.randn(rows, features, device='cuda').to(dtype)
idx = torch.randint(0, rows, (proj, ), device='cuda')
if use_torch:
    linear.use_native_sketch(False)
    gen = torch.Generator(device='cuda').manual_seed(1)

    def call():
        return linear._sketch('dct', m, proj, gen)
else:
    ws = torch.empty(cabi.sampled_dct_workspace_bytes(rows, features, proj, dtype), dtype=torch.uint8, device='cuda')
    o = torch.empty(proj, features, dtype=dtype, device='cuda')

    def call():
        if mode == 'seeded':
            return cabi.sampled_dct_seeded(m, proj, 1234, 1.0, out=o, workspace=ws)
        return cabi.sampled_dct(m, idx, 1.0, out=o, workspace=ws)

settle_calls = 0
t0 = time.perf_counter()
while (time.perf_counter() - t0) * 1e3 < settle_ms:
    for _ in range(10):
        call()
    settle_calls += 10
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
call()
settle_calls += 1
e0.record()
for _ in range(reps):
    call()
e1.record()
torch.cuda.synchronize()
es = m.element_size()
floor = (rows + proj) * features * es
moved = rows * features * es + 2 * ((features + 63) // 64) * rows * 256 + proj * features * es
us = e0.elapsed_time(e1) * 1e3 / reps
print(json.dumps({'what': {'torch': 'torch.fft formulation', 'seeded': 'fewbit_hip_sampled_dct_seeded', 'explicit': 'fewbit_hip_sampled_dct'}[mode], 'rows': rows, 'features': features, 'proj': proj, 'dtype': dname,
                  'reps': reps, 'settle_ms': settle_ms, 'settle_calls': settle_calls, 'event_us_per_call': round(us, 2),
                  'byte_floor': floor, 'x_byte_floor_at_8TBs': round(us / (floor / 8e6), 2),
                  'bytes_moved_by_design': moved, 'GBs_of_bytes_moved': round(moved / us / 1e3, 1)}))
