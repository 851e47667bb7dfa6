"""warm / cold forward and step times of the build selected by FEWBIT_HIP_LIB, several configs (one line per config)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store
dev = 'cuda'
def timeit(fns, rounds):
    for f in fns: f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for f in fns: f()
    e0.record()
    for _ in range(rounds):
        for f in fns: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / rounds / len(fns)
tag = os.path.basename(os.environ.get('FEWBIT_HIP_LIB', 'default')).replace('libfewbit_hip_', '').replace('.so', '') + os.environ.get('TAGX', '')
cfgs = [('gelu', 3, torch.bfloat16, 4096 * 4096), ('gelu', 3, torch.bfloat16, 8192 * 4096), ('silu', 4, torch.float16, 8192 * 8192),
        ('silu', 2, torch.float16, 8192 * 8192)]
if len(sys.argv) > 1: cfgs = cfgs[:int(sys.argv[1])]
DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[os.environ.get('DT', 'bf16')]
if os.environ.get('SIZES'): cfgs = [('gelu', 3, DT, int(v)) for v in os.environ['SIZES'].split(',')]
for name, k, dtype, n in cfgs:
    es = torch.empty(0, dtype=dtype).element_size()
    per_set = n * (4 * es + k / 8)
    nsets = max(3, int(1.25 * 2**30 / per_set) + 1)
    bo, lv = store.get(name, k, dev, dtype); bo = bo[1:-1].contiguous()
    F, B = [], []
    keep = []
    for _ in range(nsets):
        x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x); gy = torch.randn(n, device=dev).to(dtype); gx = torch.empty_like(x)
        st = torch.empty(cabi.state_nbytes(n, k), dtype=torch.uint8, device=dev)
        F.append(cabi.bind_forward(name, x, bo, out=y, state=st)); B.append(cabi.bind_backward(gy, st, lv, out=gx)); keep.append((x, y, gy, gx, st))
    S = []
    for f, b in zip(F, B): S += [f, b]
    rc = max(3, 300 // nsets)
    res = []
    for rep in range(3):
        fw = timeit([F[0]], 500); sw = 2 * timeit([F[0], B[0]], 300); bw = timeit([B[0]], 500)
        fc = timeit(F, rc); sc = 2 * timeit(S, rc); bc = timeit(B, rc)
        res.append((fw, sw, fc, sc, bw, bc))
    fw, sw, fc, sc, bw, bc = [min(r[i] for r in res) for i in range(6)]
    fb = n * (2 * es + k / 8)
    print(f'{tag:8s} {name}{k} {str(dtype)[6:]:8s} n={n:9d}: warm fwd {fw:6.2f} us ({fb/fw/8e4:5.1f}%) step {sw:6.2f} us ({2*fb/sw/8e4:5.1f}%) | '
          f'cold fwd {fc:6.2f} us ({fb/fc/8e4:5.1f}%) step {sc:6.2f} us ({2*fb/sc/8e4:5.1f}%) | bwd warm {bw:6.2f} cold {bc:6.2f}', flush=True)
    del F, B, S, keep
    torch.cuda.empty_cache()
