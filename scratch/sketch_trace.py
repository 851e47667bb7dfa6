"""per-stage shader-clock stamps of workgroup (0,0,0) of the random-projection kernel (FEWBIT_SKETCH_TRACE build)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ['FEWBIT_HIP_LIB'] = os.path.join(ROOT, 'scratch', 'libfewbit_hip_sktrace.so')
import numpy as np, torch
from fewbit_amd import cabi
dist = sys.argv[1] if len(sys.argv) > 1 else 'rademacher'
w = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows, features, proj = 16384, 3072, 1638
cabi.tune_sketch_waves(w)
halves = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cabi.tune_sketch_halves(halves)
if halves == 2: w = 8
m = torch.randn(rows, features, device='cuda').to(torch.bfloat16)
plan = cabi.describe_sketch(dist, rows, features, proj)
for _ in range(3): cabi.sketch(dist, m, proj, 1, 1.0)
torch.cuda.synchronize()
L = cabi.lib()
L.fewbit_hip_sketch_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(8 * 512 * 12, dtype=np.uint64)
assert L.fewbit_hip_sketch_debug_trace(buf.ctypes.data, buf.size) == 0
t = buf.reshape(8, 512, 12).astype(np.int64)
nst = plan['k_slice'] // (16 * w // halves)
print(plan, 'stages per slice', nst)
for wave in range(w):
    tt = t[wave, :nst]
    mf = tt[:, 1] - tt[:, 0]; bar = tt[:, 2] - tt[:, 1]; tot = np.diff(tt[:, 0])
    print('wave %d: stage total median %d cycles (p10 %d, p90 %d); top->MFMAs issued median %d; barrier wait median %d (p90 %d); whole slice %d cycles'
          % (wave, np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90), np.median(mf), np.median(bar), np.percentile(bar, 90), tt[-1, 2] - tt[0, 0]))
ksteps = (16 * w // halves) // 16
for wave in (0, w - 1):
    tt = t[wave, 2:nst - 2]
    marks = np.concatenate([tt[:, 3:3 + ksteps], tt[:, 1:2]], axis=1)
    print('wave %d: median cycles per MFMA step (8 MFMAs each):' % wave, [int(x) for x in np.median(np.diff(marks, axis=1), axis=0)], ' top->first step', int(np.median(tt[:, 3] - tt[:, 0])))
print('first 12 stages of wave 0 (total, mfma-section, barrier):', [(int(a), int(b), int(c)) for a, b, c in zip(np.diff(t[0, :13, 0]), (t[0, :12, 1] - t[0, :12, 0]), (t[0, :12, 2] - t[0, :12, 1]))])
