"""A/B of several builds (and/or tune settings) of libfewbit_hip in ONE process, interleaved round by round (robust against
clock drift):
    python scratch/ablate.py fwd|bwd|step name=path[@key=value,key=value] ...   (env SIZE=elements, K=bits, DT=bf16|f16|f32, FN=gelu,
                                                                                  INPLACE=1: outputs alias inputs)
Every library is loaded with ctypes directly; times are medians over rounds of 200 back-to-back launches, warm (one
buffer set) and cold (rotating sets, > 1.25 GiB)."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store

which = sys.argv[1]
libs = [a.split('=', 1) for a in sys.argv[2:]]
KEYS = ('waves_per_cu', 'chunk', 'lut_chunk', 'lut_blocks_per_cu', 'lut_min', 'lut_block', 'u_fwd', 'u_bwd', 'u_lut', 'u_step1')
n = int(os.environ.get('SIZE', 4096 * 4096)); k = int(os.environ.get('K', 3)); fn = os.environ.get('FN', 'gelu')
dtype = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[os.environ.get('DT', 'bf16')]
tune = dict(kv.split('=') for kv in os.environ.get('TUNE', '').split(',') if kv)
inplace = os.environ.get('INPLACE', '0') == '1'     # y aliases x, gx aliases gy (the reference operator's own mode)
dev = 'cuda'
es = torch.empty(0, dtype=dtype).element_size()
fb = n * (2 * es + k / 8)
nsets = max(3, int(1.25 * 2**30 / (n * (4 * es + k / 8))) + 1)
bo, lv = store.get(fn, k, dev, dtype); bo = bo[1:-1].contiguous()
sets = []
for _ in range(nsets):
    x = torch.randn(n, device=dev).to(dtype); sets.append((x, torch.empty_like(x), torch.randn(n, device=dev).to(dtype), torch.empty_like(x),
                                                          torch.empty(cabi.state_nbytes(n, k), dtype=torch.uint8, device=dev)))
stream = torch.cuda.current_stream().cuda_stream
vp, sz, i32, dbl = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_double
L = {}
SET = {}
for name, path in libs:
    path, _, own = path.partition('@')
    own = dict(kv.split('=') for kv in own.split(',') if kv)
    lib = ctypes.CDLL(os.path.abspath(path))
    SET[name] = (lib, dict(tune, **own))
    lib.fewbit_hip_quantize_forward.argtypes = [i32, i32, vp, vp, vp, sz, vp, i32, dbl, dbl, vp]
    lib.fewbit_hip_quantize_backward.argtypes = [i32, vp, vp, vp, sz, vp, i32, vp]
    if hasattr(lib, 'fewbit_hip_tune'): lib.fewbit_hip_tune.argtypes = [ctypes.c_char_p, ctypes.c_longlong]
    calls = []
    for (x, y, gy, gx, st) in sets:
        af = (cabi.CONTINUOUS.index(fn), cabi.DTYPES[dtype], x.data_ptr(), (x if inplace else y).data_ptr(), st.data_ptr(), n, bo.data_ptr(), bo.numel(), 0.0, 0.0, stream)
        ab = (cabi.DTYPES[dtype], gy.data_ptr(), st.data_ptr(), (gy if inplace else gx).data_ptr(), n, lv.data_ptr(), lv.numel(), stream)
        if which in ('fwd', 'step'):
            calls.append((lib.fewbit_hip_quantize_forward, af))
        if which in ('bwd', 'step'):
            calls.append((lib.fewbit_hip_quantize_backward, ab))
    L[name] = calls
# a valid state for the backward
f0 = cabi.lib()
for (x, y, gy, gx, st) in sets: cabi.quantize_forward(fn, x, bo, out=y, state=st)
torch.cuda.synchronize()

def apply(name):
    lib, settings = SET[name]
    if not hasattr(lib, 'fewbit_hip_tune'): return        # an older build (no run-time tuning)
    for kk in KEYS: lib.fewbit_hip_tune(kk.encode(), -1)
    for kk, vv in settings.items(): assert lib.fewbit_hip_tune(kk.encode(), int(vv)) == 0, kk

def run(calls, reps):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for f, a in calls[:2]: assert f(*a) == 0
    e0.record()
    for _ in range(reps):
        for f, a in calls: f(*a)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / len(calls)

a = torch.empty(1 << 26, device=dev)
for _ in range(300): a.add_(1.0)
torch.cuda.synchronize()
warm = {nm: [] for nm in L}; cold = {nm: [] for nm in L}
for r in range(int(os.environ.get('ROUNDS', 7))):
    for nm, calls in L.items():
        apply(nm)
        warm[nm].append(run(calls[:2] if which == 'step' else calls[:1], 200))
    for nm, calls in L.items():
        apply(nm)
        cold[nm].append(run(calls, max(2, 100 // nsets)))
print(f'# {which}{" IN PLACE" if inplace else ""} {fn} k={k} {str(dtype)[6:]} n={n} tune={tune}: median us per launch over {len(next(iter(warm.values())))} interleaved rounds (min..max); % of 8 TB/s on {fb:.0f} B')
for nm in L:
    w, c = statistics.median(warm[nm]), statistics.median(cold[nm])
    print(f'{nm:14s} warm {w:6.2f} ({min(warm[nm]):6.2f}..{max(warm[nm]):6.2f}) {fb/w/8e4:5.1f}%   cold {c:6.2f} ({min(cold[nm]):6.2f}..{max(cold[nm]):6.2f}) {fb/c/8e4:5.1f}%', flush=True)
