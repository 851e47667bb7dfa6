"""Launch-shape sweep of the streaming kernels inside ONE process (fewbit_hip_tune): groups per lane per stage (U) x resident
blocks per CU x resident/chunked shape, warm (one buffer set) and cold (rotating > 1.25 GiB), for the BASELINE sizes.

    FEWBIT_HIP_LIB=scratch/libfewbit_hip_sweep.so python scratch/shape_sweep.py [bwd|lut|search|step1|all] [c2,c4,c3k2,c3k4,f32]

The sweep build (make -C fewbit_amd/csrc variant NAME=sweep DEFS=-DFEWBIT_SWEEP) holds every U and both table block sizes for
gelu / silu / relu only.  One line per setting; `*` marks the built-in policy's own setting."""
import itertools
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store

dev = 'cuda'
what = sys.argv[1] if len(sys.argv) > 1 else 'all'
names = sys.argv[2].split(',') if len(sys.argv) > 2 else ['c2', 'c4', 'c3k2', 'c3k4', 'f32']
CFG = {'c2': ('gelu', 3, torch.bfloat16, 4096 * 4096), 'c4': ('gelu', 3, torch.bfloat16, 8192 * 4096),
       'c3k2': ('silu', 2, torch.float16, 8192 * 8192), 'c3k4': ('silu', 4, torch.float16, 8192 * 8192),
       'f32': ('gelu', 3, torch.float32, 4096 * 4096), 'c1': ('relu', 1, torch.float32, 1024 * 1024),
       'rob': ('gelu', 3, torch.float32, 16384 * 3072), 'robbf': ('gelu', 3, torch.bfloat16, 16384 * 3072),
       'relu16': ('relu', 1, torch.bfloat16, 8192 * 4096), 'relu16c2': ('relu', 1, torch.bfloat16, 4096 * 4096),
       'relu32': ('relu', 1, torch.float32, 16384 * 3072), 'relu16_2m': ('relu', 1, torch.bfloat16, 2 << 20),
       'relu16_4m': ('relu', 1, torch.bfloat16, 4 << 20), 'relu16_8m': ('relu', 1, torch.bfloat16, 8 << 20),
       'relu16_12m': ('relu', 1, torch.bfloat16, 12 << 20), 'relu16_25m': ('relu', 1, torch.bfloat16, 25 << 20),
       'relu32_16m': ('relu', 1, torch.float32, 16 << 20), 'relu32_8m': ('relu', 1, torch.float32, 8 << 20),
       'f32_8m': ('gelu', 3, torch.float32, 8 << 20), 'f32_4m': ('gelu', 3, torch.float32, 4 << 20), 'bf16_8m': ('gelu', 3, torch.bfloat16, 8 << 20),
       'bf16_12m': ('gelu', 3, torch.bfloat16, 12 << 20), 'bf16_4m': ('gelu', 3, torch.bfloat16, 4 << 20)}
ALL = ('waves_per_cu', 'chunk', 'lut_chunk', 'lut_blocks_per_cu', 'lut_min', 'lut_block', 'u_fwd', 'u_bwd', 'u_lut', 'u_step1')


def reset():
    cabi.tune(**{k: -1 for k in ALL})


def timeit(fns, rounds):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for f in fns: f()
    e0.record()
    for _ in range(rounds):
        for f in fns: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / rounds / len(fns)


def best(fns, rounds, reps=3):
    return min(timeit(fns, rounds) for _ in range(reps))


class Sets:
    def __init__(self, name, k, dtype, n):
        es = torch.empty(0, dtype=dtype).element_size()
        per_set = n * (4 * es + k / 8)
        self.nsets = max(3, int(1.25 * 2**30 / per_set) + 1)
        self.fb = n * (2 * es + k / 8)
        self.F, self.B, self.keep = [], [], []
        step1 = name == 'relu'
        if not step1:
            bo, lv = store.get(name, k, dev, dtype); bo = bo[1:-1].contiguous()
        for _ in range(self.nsets):
            x = torch.randn(n, device=dev).to(dtype); y = torch.empty_like(x); gy = torch.randn(n, device=dev).to(dtype); gx = torch.empty_like(x)
            st = torch.empty(cabi.state_nbytes(n, k), dtype=torch.uint8, device=dev)
            if step1:
                self.F.append(cabi.bind_stepwise1_forward(name, x, out=y, state=st)); self.B.append(cabi.bind_stepwise1_backward(name, gy, st, out=gx))
            else:
                self.F.append(cabi.bind_forward(name, x, bo, out=y, state=st)); self.B.append(cabi.bind_backward(gy, st, lv, out=gx))
            self.keep.append((x, y, gy, gx, st))
        for f, b in zip(self.F, self.B): f(); b()
        torch.cuda.synchronize()

    def measure(self, which):
        L = self.F if which == 'fwd' else self.B
        warm = best([L[0]], 300)
        cold = best(L, max(3, 200 // self.nsets))
        return warm, cold


def settle():
    a = torch.empty(1 << 26, device=dev)
    for _ in range(200): a.add_(1.0)
    torch.cuda.synchronize()


def run(cfgname, kind):
    name, k, dtype, n = CFG[cfgname]
    S = Sets(name, k, dtype, n)
    fb = S.fb
    settle()
    rows = []
    if kind == 'bwd':
        key, which = 'u_bwd', 'bwd'
        grid = [dict(u_bwd=u, waves_per_cu=w, chunk=c) for u in (1, 2, 4) for w in (8, 16, 24, 32) for c in (0, 1, 2, 3)]
        desc = lambda: cabi.describe_backward(dtype, n, 2**k)
    elif kind == 'search':
        which = 'fwd'
        grid = [dict(lut_min=1 << 60, u_fwd=u, waves_per_cu=w, chunk=c) for u in (1, 2, 4) for w in (8, 16, 24, 32) for c in (0, 1, 3)]
        desc = lambda: cabi.describe_forward(name, dtype, n, 2**k - 1)
    elif kind == 'lut':
        which = 'fwd'
        grid = [dict(lut_min=0, lut_block=b, u_lut=u, lut_blocks_per_cu=p, lut_chunk=c) for b in (1024, 512) for u in (1, 2, 4) for p in (2, 1)
                for c in (0, 1, 2, 3)]
        desc = lambda: cabi.describe_forward(name, dtype, n, 2**k - 1)
    elif kind == 'step1f':
        which = 'fwd'
        grid = [dict(u_step1=u, waves_per_cu=w, chunk=c) for u in (1, 2, 4) for w in (8, 16, 32) for c in (0, 1, 3)]
        desc = lambda: cabi.describe_stepwise1_forward(name, dtype, n)
    else:  # step1b
        which = 'bwd'
        grid = [dict(u_step1=u, waves_per_cu=w, chunk=c) for u in (1, 2, 4) for w in (8, 16, 32) for c in (0, 1)]
        desc = lambda: cabi.describe_stepwise1_backward(name, dtype, n)
    reset()
    d0 = desc()
    w0, c0 = S.measure(which)
    print(f'## {cfgname} {kind}: {name} k={k} {str(dtype)[6:]} n={n}  algorithmic bytes {fb:.0f}; policy: {d0["kernel"]} blocks={d0["blocks"]} '
          f'({d0["blocks_per_cu"]}/CU) chunk={d0["chunk"]}: warm {w0:.2f} us ({fb/w0/8e4:.1f}%) cold {c0:.2f} us ({fb/c0/8e4:.1f}%)', flush=True)
    seen = set()
    for g in grid:
        reset(); cabi.tune(**g)
        d = desc()
        sig = (d['kernel'], d['blocks'], d['chunk'])
        if sig in seen: continue
        seen.add(sig)
        try:
            w, c = S.measure(which)
        except Exception as e:  # noqa
            print('   failed', g, e); continue
        tag = ' '.join(f'{k_}={v}' for k_, v in g.items() if k_ != 'lut_min')
        rows.append((w, c, tag, d))
        print(f'   {tag:52s} blocks={d["blocks"]:6d} ({d["blocks_per_cu"]}/CU x{d["threads"]}) chunk={d["chunk"]} U={d["u"]}: '
              f'warm {w:6.2f} us ({fb/w/8e4:5.1f}%)  cold {c:6.2f} us ({fb/c/8e4:5.1f}%)', flush=True)
    rows.sort(key=lambda r: r[0])
    print('   best warm:', '; '.join(f'{r[2]} -> {r[0]:.2f}' for r in rows[:3]))
    rows.sort(key=lambda r: r[1])
    print('   best cold:', '; '.join(f'{r[2]} -> {r[1]:.2f}' for r in rows[:3]), flush=True)
    reset()
    del S
    torch.cuda.empty_cache()


kinds = ['bwd', 'lut', 'search'] if what == 'all' else what.split(',')
for cfgname in names:
    for kind in kinds:
        if CFG[cfgname][0] == 'relu' and kind in ('bwd', 'lut', 'search'): continue
        if CFG[cfgname][0] != 'relu' and kind.startswith('step1'): continue
        if kind == 'lut' and CFG[cfgname][2] == torch.float32: continue
        run(cfgname, kind)
