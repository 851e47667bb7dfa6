"""One configuration of the random-projection kernel -- the program rocprofv3 wraps (tools/profile_sketch.sh).

    python3 tools/sketch_run.py <dist> <rows> <features> <proj> <bf16|f16|f32> [reps=200] [settle_ms=40]

The GPU is kept busy with the same launches for `settle_ms` first (an idle GPU boosts for ~1.5 ms, then runs 8-12 % slower for
~10 ms, then settles: profiles/r05_clock_transient_timeline.txt), exactly as bench.py settles its own figures; then `reps`
launches are timed between two HIP events.  Prints one JSON line: the plan, how many launches preceded the timed ones (the
summary script drops that many dispatches per kernel from the trace) and the event time of the timed launches.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fewbit_amd import cabi  # noqa: E402

dist = sys.argv[1] if len(sys.argv) > 1 else 'rademacher'
rows, features, proj = (int(a) for a in (sys.argv[2:5] if len(sys.argv) > 4 else (16384, 3072, 3276)))
dname = sys.argv[5] if len(sys.argv) > 5 else 'bf16'
dtype = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f16': torch.float16}[dname]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 200
settle_ms = float(sys.argv[7]) if len(sys.argv) > 7 else 40.0

m = torch.randn(rows, features, device='cuda').to(dtype)
plan = cabi.describe_sketch(dist, rows, features, proj, dtype)
ws = torch.empty(max(plan['workspace_bytes'], 1), dtype=torch.uint8, device='cuda')
o = torch.empty(proj, features, dtype=dtype, device='cuda')


def call():
    cabi.sketch(dist, m, proj, 1234, 1.0 / proj, out=o, workspace=ws)


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
print(json.dumps({'dist': dist, 'rows': rows, 'features': features, 'proj': proj, 'dtype': dname, 'reps': reps, 'settle_ms': settle_ms,
                  'settle_calls': settle_calls, 'event_us_per_call': round(e0.elapsed_time(e1) * 1e3 / reps, 2), 'plan': plan}))
