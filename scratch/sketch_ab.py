"""A/B of builds / tile heights of the random-projection kernel in ONE process, interleaved round by round:
   python scratch/sketch_ab.py dist rows features proj name=lib[@waves=W,slices=Z,halves=H,partials=P,mem=M] ..."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
dist, rows, features, proj = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dtype = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f16': torch.float16}[os.environ.get('DT', 'bf16')]
m = torch.randn(rows, features, device='cuda').to(dtype)
o = torch.empty(proj, features, dtype=dtype, device='cuda')
ws = torch.empty(proj * features * 4 * 16 + (proj + 256) * (rows + 512) * 2 + (rows * features * 2 if dtype == torch.float32 else 0) + 65536, dtype=torch.uint8, device='cuda')
vp, sz, i32, dbl, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_double, ctypes.c_uint64
arms = {}
for a in sys.argv[5:]:
    name, spec = a.split('=', 1)
    path, _, own = spec.partition('@')
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.fewbit_hip_sketch.argtypes = [i32, i32, vp, sz, sz, sz, sz, u64, dbl, vp, vp, sz, vp]
    lib.fewbit_hip_tune.argtypes = [ctypes.c_char_p, ctypes.c_longlong]       # (ABI 5: the sketch settings are keys of the one hook)
    arms[name] = (lib, dict(kv.split('=') for kv in own.split(',') if kv))
stream = torch.cuda.current_stream().cuda_stream
def run(name, reps):
    lib, st = arms[name]
    for key, name_ in (('slices', 'sketch_slices'), ('waves', 'sketch_waves'), ('halves', 'sketch_halves'), ('partials', 'sketch_partials'), ('mem', 'sketch_materialise')):
        lib.fewbit_hip_tune(name_.encode(), int(st.get(key, -1)))
    args = (cabi.SKETCH_DISTS.index(dist), cabi.DTYPES[dtype], m.data_ptr(), rows, features, features, proj, 1234, 1.0 / proj, o.data_ptr(), ws.data_ptr(), ws.numel(), stream)
    for _ in range(2): assert lib.fewbit_hip_sketch(*args) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.fewbit_hip_sketch(*args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for name in arms: run(name, 5)
res = {n: [] for n in arms}
for r in range(int(os.environ.get('ROUNDS', 7))):
    for n in arms: res[n].append(run(n, 20))
fl = 2.0 * proj * rows * features
print(f'# {dist} {rows}x{features} proj {proj} {str(dtype)[6:]}: median us over {len(next(iter(res.values())))} interleaved rounds of 20 launches (min..max), TFLOP/s')
for n, v in res.items():
    md = statistics.median(v)
    print(f'{n:18s} {md:7.1f} ({min(v):7.1f}..{max(v):7.1f})  {fl / md / 1e6:6.0f} TFLOP/s')
