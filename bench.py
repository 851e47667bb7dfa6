#!/usr/bin/env python3
"""bench.py -- forward+backward throughput of the few-bit activation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c4|c4_tensor] [--scaling weak|strong]
                    [--digests] [--no-extras] [--no-cpu-baseline] [--no-pmc]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config c4]

One "step" = one pass of the hot path over one batch: the fused quantize+pack forward and the fused unpack+mul
backward of fewbit.gelu(bits=3), both through the C-ABI (include/fewbit_hip.h) on HBM-resident synthetic tensors.
  --config c2 (default)  4096x4096 bf16 per GPU              (BASELINE.json configs[1], the headline metric)
  --config c4            8192x4096 bf16 per GPU = the shard one GPU owns of BASELINE.json configs[3]
                         (4 x (16384x4096) bf16 over 8 GPUs, cut by fewbit_amd.sharding.shard_range)

  --config c4_tensor     16384x4096 bf16 = ONE whole tensor of configs[3] (meant for --scaling strong)

N > 1, --scaling weak (default, the driver's contract): every rank runs the same per-GPU workload on its own shard.
N > 1, --scaling strong: ONE tensor of the config's size (north_star's literal "a 4096x4096 bf16 tensor at 1/2/4/8 GPUs")
is cut by fewbit_amd.sharding.shard_range over the N ranks; rank r holds exactly elements [begin_r, end_r) of the seeded
global tensor and runs the path on that slice; `value` = bytes of the WHOLE tensor / the slowest rank's time.  At N = 8
a rank has 2 Mi elements (4 MiB in, ~4 us kernels): that line measures the launch boundary, and says so.
The path has NO exchange step, so no collective is involved anywhere.  Two ways to start it, same worker code:
  * `python bench.py --gpus N` by itself: this process starts N fresh children (one per device, the reference's own
    model: benchmark/README.md:18, benchmark/benchmark.py:149-162), BEFORE it touches the GPU; the barriers around the
    timed region are files in a private temporary directory; the parent prints the one JSON line.  A failed rank makes
    the parent exit non-zero.  Children are never exec'ed over a process that initialised the GPU.
  * under torch.distributed.run: torch.distributed carries the barrier and the max over ranks only -- over gloo (CPU)
    by default, because there is nothing for RCCL to move (FEWBIT_BENCH_BACKEND=nccl uses RCCL for the same two calls).
Ranks take device `local_rank % device_count`, so more ranks than GPUs share devices (a validation hook for 1-GPU
boxes; the line then says "shared_gpu": true and prices the roofline against the devices actually used).  When there
are at least as many devices as ranks the ranks' devices must be pairwise distinct, or the run fails.
A rank that raises writes its traceback to `<sync dir>/error.<rank>`; the parent prints it and exits non-zero.
`--digests` adds per-rank SHA-256 digests of (y, state, gx) to the line (computed after the timed region): the
strong-scaling test compares them with slices of the unsharded result (tests/test_gpu_bench.py).

Timed region (the driver's contract): exactly W warm-up steps, barrier + synchronize, host clock, EXACTLY K steps,
synchronize, host clock, barrier.  `value` / `ms_per_step` = wall time of those K steps, max over ranks.  A second,
identical K-step region between two HIP events gives the GPU-side time of such a region (`timing.event_ms_per_step`; it
excludes the first launch's latency from an idle queue and the wake-up after the final synchronize, ~18 us = 4 % of a
20-step region).
`--settle-ms T` (default 0 = off) extends the warm-up until the GPU has been busy T ms (clock transient after idle).

metric = algorithmic bytes / time; algorithmic bytes per element = 4*s + k/4 (fwd: read x, write y, write state;
bwd: read gy, read state, write gx; s = element size, k = bits) -- SURVEY.md 8(d).

Besides the contract fields the line carries (N = 1, unless --no-extras):
  roofline   dominant kernel (name from fewbit_hip_describe_*), its algorithmic GB/s against the 8 TB/s HBM peak; `traffic` =
             HBM bytes per launch from two rocprofv3 --pmc child runs of this very command (measure_traffic; --no-pmc skips them
             and the figure recorded under profiles/ is replayed, labelled so)
  cold       the same workload rotating through > 1 GiB of independent buffer sets (Infinity Cache out of the picture)
  configs    every other BASELINE config on one GPU, warm and cold, with kernel names and its own cpu_baseline
  op_level   clone + torch.ops.fewbit.gelu + autograd.grad at this size beside torch.nn.functional.gelu
  sketch     the random-projection kernel of the randomized linear layers at RoBERTa's widest layer: MFMA roofline
  cpu_baseline, cpu_baseline_1thread   the reference's own CPU path (oracle/_ref) on the host cores, bounded samples
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
INFINITY_CACHE_BYTES = 256 << 20
DTYPES = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}

# BASELINE.json configs.  `kind`: continuous = k-bit table kernels, stepwise1 = exact 1-bit family
CONFIGS = {
    'c2': dict(fn='gelu', bits=3, rows=4096, cols=4096, dtype='bf16', kind='continuous',
               label='fewbit.gelu bits=3 on 4096x4096 bf16'),
    'c4': dict(fn='gelu', bits=3, rows=8192, cols=4096, dtype='bf16', kind='continuous',
               label='fewbit.gelu bits=3 on 8192x4096 bf16 = one GPU\'s shard of 4x(16384x4096) over 8 GPUs'),
    'c4_tensor': dict(fn='gelu', bits=3, rows=16384, cols=4096, dtype='bf16', kind='continuous',
                      label='fewbit.gelu bits=3 on 16384x4096 bf16 = one whole tensor of 4x(16384x4096)'),
    'c1': dict(fn='relu', bits=1, rows=1024, cols=1024, dtype='f32', kind='stepwise1',
               label='fewbit.relu 1-bit on 1024x1024 fp32'),
    'c3_k2': dict(fn='silu', bits=2, rows=8192, cols=8192, dtype='f16', kind='continuous',
                  label='fewbit.silu bits=2 on 8192x8192 fp16'),
    'c3_k4': dict(fn='silu', bits=4, rows=8192, cols=8192, dtype='f16', kind='continuous',
                  label='fewbit.silu bits=4 on 8192x8192 fp16'),
    'c2_fp32': dict(fn='gelu', bits=3, rows=4096, cols=4096, dtype='f32', kind='continuous',
                    label='fewbit.gelu bits=3 on 4096x4096 fp32'),
    # the only op-level timing the reference publishes (notebooks/few-bit-backward/
    # memory-usage-operation-only.py:41,70,80-85 -- torch.ops.fewbit.gelu with 3-bit tables on 128*2^20 fp32 elements,
    # forward 2.862 + backward 1.563 = 4.425, unit assumed ms, GPU not stated => ~473 GiB/s in this metric)
    'ref_notebook': dict(fn='gelu', bits=3, rows=131072, cols=1024, dtype='f32', kind='continuous',
                         label='fewbit.gelu bits=3 on 128*2^20 fp32 elements (shape of the reference notebook timing)',
                         reference_published={'fwd_plus_bwd': 4.425, 'unit': 'ms (assumed; not stated)', 'GiB_s': 473.0,
                                              'hardware': 'not stated (an NVIDIA GPU)',
                                              'source': 'notebooks/few-bit-backward/memory-usage-operation-only.py:41,70,80-85'}),
}
# bounded CPU-baseline samples per config: (repetitions at all threads, repetitions at one thread); ~25 s of CPU in all
CPU_SAMPLES = {'c2': (12, 6), 'c4': (5, 0), 'c4_tensor': (3, 0), 'c1': (40, 0), 'c3_k2': (3, 0), 'c3_k4': (3, 0), 'c2_fp32': (6, 0)}


def load_tables(cfg, device):
    dt = DTYPES[cfg['dtype']]
    with np.load(ROOT / 'fewbit_amd' / 'data' / 'builtin.npz') as z:
        key = f"{cfg['fn']}{cfg['bits']:02d}"
        borders = torch.tensor(z[f'{key}-borders']).to(dt)[1:-1].contiguous().to(device)
        levels = torch.tensor(z[f'{key}-levels']).to(dt).to(device)
    return borders, levels


def step_bytes(cfg, n=None):
    """(fwd+bwd, fwd only) algorithmic bytes of one step over `n` elements (default: the whole config tensor)"""
    es = torch.empty(0, dtype=DTYPES[cfg['dtype']]).element_size()
    n = cfg['rows'] * cfg['cols'] if n is None else n
    return n * (4 * es + cfg['bits'] / 4), n * (2 * es + cfg['bits'] / 8)


def describe(cfg, n=None):
    """kernel instantiation + launch shape the library uses for this config on the current device (nothing is launched)"""
    from fewbit_amd import cabi
    dt, n = DTYPES[cfg['dtype']], (cfg['rows'] * cfg['cols'] if n is None else n)
    if cfg['kind'] == 'continuous':
        return (cabi.describe_forward(cfg['fn'], dt, n, 2 ** cfg['bits'] - 1), cabi.describe_backward(dt, n, 2 ** cfg['bits']))
    return cabi.describe_stepwise1_forward(cfg['fn'], dt, n), cabi.describe_stepwise1_backward(cfg['fn'], dt, n)


class Workload:
    """`nsets` independent buffer sets (x, y, gy, gx, state) of one config with pre-resolved launches per set.
    `span` = (begin, end): this workload is the slice [begin, end) of the config's flattened tensor (strong scaling);
    with host seeding the slice holds exactly those elements of the seeded global tensor."""

    def __init__(self, cfg, device, nsets=1, seed=0, host_seeded=True, span=None):
        from fewbit_amd import cabi
        dt = DTYPES[cfg['dtype']]
        total = cfg['rows'] * cfg['cols']
        begin, end = span if span is not None else (0, total)
        n = end - begin
        shape = (cfg['rows'], cfg['cols']) if span is None else (n,)
        self.cfg, self.n, self.nsets, self.span = cfg, n, nsets, (begin, end)
        self.fwd, self.bwd, self.keep = [], [], []
        if cfg['kind'] == 'continuous':
            borders, levels = load_tables(cfg, device)
        for i in range(nsets):
            if host_seeded:     # synthetic shard: seeded on the host, then resident in HBM
                g = torch.Generator().manual_seed(2 * seed + 1000 * i)
                x = torch.randn(total, generator=g).to(dt)[begin:end].reshape(shape).contiguous().to(device)
                g = torch.Generator().manual_seed(2 * seed + 1 + 1000 * i)
                gy = torch.randn(total, generator=g).to(dt)[begin:end].reshape(shape).contiguous().to(device)
            else:               # the extra measurements: same distribution, drawn on the device
                g = torch.Generator(device=device).manual_seed(2 * seed + 1000 * i)
                x = torch.randn(shape, generator=g, device=device).to(dt)
                gy = torch.randn(shape, generator=g, device=device).to(dt)
            y, gx = torch.empty_like(x), torch.empty_like(x)
            state = torch.empty(cabi.state_nbytes(n, cfg['bits']), dtype=torch.uint8, device=device)
            if cfg['kind'] == 'continuous':
                self.fwd.append(cabi.bind_forward(cfg['fn'], x, borders, out=y, state=state))
                self.bwd.append(cabi.bind_backward(gy, state, levels, out=gx))
            else:
                self.fwd.append(cabi.bind_stepwise1_forward(cfg['fn'], x, out=y, state=state))
                self.bwd.append(cabi.bind_stepwise1_backward(cfg['fn'], gy, state, out=gx))
            self.keep.append((x, y, gy, gx, state))
        self.set_bytes = sum(t.numel() * t.element_size() for t in self.keep[0])

    def steps(self):
        """launch list of one rotation: fwd(set 0), bwd(set 0), fwd(set 1), ..."""
        out = []
        for f, b in zip(self.fwd, self.bwd):
            out += [f, b]
        return out

    def digests(self):
        """SHA-256 of y, state, gx of set 0 as they stand (after at least one step)"""
        import hashlib
        torch.cuda.synchronize()
        _, y, _, gx, state = self.keep[0]
        return {name: hashlib.sha256(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()
                for name, t in (('y', y), ('state', state), ('gx', gx))}


def event_time_us(launches, rounds, preroll=1):
    """Average GPU time of one pass over `launches`, between two events on the launch stream; the queue is filled by
    `preroll` untimed passes first so the first timed launch does not wait for the host."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    for _ in range(preroll):
        for f in launches:
            f()
    e0.record()
    for _ in range(rounds):
        for f in launches:
            f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / rounds


def nsets_for_cold(cfg):
    """buffer sets so that a rotation touches > 1 GiB and > 4x the Infinity Cache between two uses of a set"""
    es = torch.empty(0, dtype=DTYPES[cfg['dtype']]).element_size()
    n = cfg['rows'] * cfg['cols']
    per_set = n * (4 * es + cfg['bits'] / 8)
    return max(3, int(max(1.25 * 2**30, 4.5 * INFINITY_CACHE_BYTES) / per_set) + 1)


def measure_config(cfg, device, cold):
    """us_fwd / us_bwd: K back-to-back launches of one kernel (rotating over the sets when cold); us_step: fwd+bwd
    alternating, which is what the headline metric is defined on."""
    nsets = nsets_for_cold(cfg) if cold else 1
    w = Workload(cfg, device, nsets=nsets, seed=7, host_seeded=False)
    total = max(200, 3 * nsets)                        # launches per measurement
    rounds = max(1, total // nsets)
    for f in w.steps():                                # touch everything once (first-use page mapping, table casts)
        f()
    us_fwd = event_time_us(w.fwd, rounds) / nsets
    us_bwd = event_time_us(w.bwd, rounds) / nsets
    us_step = event_time_us(w.steps(), rounds) / nsets
    sb, _ = step_bytes(cfg)
    out = {'us_fwd': round(us_fwd, 2), 'us_bwd': round(us_bwd, 2), 'us_step': round(us_step, 2),
           'GiB_s': round(sb / (us_step * 1e-6) / 2**30, 1), 'frac': round(sb / (us_step * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
           'buffer_sets': nsets, 'footprint_MiB': round(nsets * w.set_bytes / 2**20, 1)}
    del w
    torch.cuda.empty_cache()
    return out


def graph_step_us(cfg, device, steps=50, replays=20):
    """GPU-side time of one fwd+bwd step with the host out of the picture: `steps` steps captured into ONE hipGraph on a side
    stream and replayed.  For tensors this small the eager loop measures the host's launch rate (~2.7 us per launch even
    from C++, profiles/r03_stream_bench_4MiB.txt), not the kernels."""
    side = torch.cuda.Stream(device)
    with torch.cuda.stream(side):
        w = Workload(cfg, device, nsets=1, seed=7, host_seeded=False)       # launches bound to the side stream
        for f in w.steps():
            f()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(steps):
            for f in w.steps():
                f()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / replays / steps * 1e6
    del g, w
    return us


def measure_op_level(cfg, device, reps=200):
    """The operator path a model sees: `clone` (the op works in place) + torch.ops.fewbit.<fn> + autograd.grad, beside the same
    three calls with torch.nn.functional.<fn> (kernel-only AND op-level numbers).  Eager (GPU time between two
    events, and host wall time per iteration: when the two agree the loop is HOST-bound -- dispatcher + autograd engine, not
    kernels) and as a hipGraph replay (the GPU-side time of the same launches with the host out of the picture)."""
    import torch.nn.functional as F
    import fewbit_amd
    if not fewbit_amd.native_loaded():
        return {'error': fewbit_amd.native_error()}
    dt = DTYPES[cfg['dtype']]
    borders, levels = load_tables(cfg, device)
    x = torch.randn(cfg['rows'], cfg['cols'], device=device).to(dt).requires_grad_()
    gy = torch.randn(cfg['rows'], cfg['cols'], device=device).to(dt)
    op = getattr(torch.ops.fewbit, cfg['fn']).default
    ref = getattr(F, cfg['fn'])

    def fewbit_step():
        return torch.autograd.grad(op(x.clone(), borders, levels), x, gy)

    def vanilla_step():
        return torch.autograd.grad(ref(x.clone()), x, gy)

    def clone_only():
        return x.detach().clone()

    out = {}
    for name, f in (('fewbit', fewbit_step), ('vanilla', vanilla_step), ('clone_alone', clone_only)):
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e6
        out[name] = {'us_gpu': round(e0.elapsed_time(e1) * 1e3 / reps, 2), 'us_wall': round(wall, 2)}
        try:                                            # the same launches captured once and replayed
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    f()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                keep = f()                              # noqa: F841  (outputs live in the graph's pool)
            for _ in range(10):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                g.replay()
            torch.cuda.synchronize()
            out[name]['us_graph_replay'] = round((time.perf_counter() - t0) / reps * 1e6, 2)
            del g, keep
        except Exception as e:  # noqa: BLE001
            out[name]['us_graph_replay'] = None
            out[name]['graph_error'] = f'{type(e).__name__}: {e}'[:200]
    sb, _ = step_bytes(cfg)
    key = 'us_graph_replay' if out['fewbit'].get('us_graph_replay') and out['clone_alone'].get('us_graph_replay') else 'us_gpu'
    net = out['fewbit'][key] - out['clone_alone'][key]
    out['fewbit_minus_clone'] = {'from': key, 'us': round(net, 2), 'GiB_s': round(sb / (net * 1e-6) / 2**30, 1),
                                 'frac': round(sb / (net * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
    out['speedup_vs_vanilla'] = {k: round(out['vanilla'][k] / out['fewbit'][k], 3) for k in ('us_gpu', 'us_graph_replay')
                                 if out['vanilla'].get(k) and out['fewbit'].get(k)}
    out['note'] = (f"{reps} iterations of x.clone() + torch.ops.fewbit.{cfg['fn']}(clone, borders, levels) + torch.autograd.grad on "
                   f"{cfg['rows']}x{cfg['cols']} {cfg['dtype']}; `vanilla` = the same with torch.nn.functional.{cfg['fn']}; us_gpu: between two "
                   "events on the stream, eager; us_wall: host time per eager iteration (us_wall ~ us_gpu => host-bound); us_graph_replay: "
                   "one hipGraph replay of the same launches (three kernels + their boundaries, ~1.8 us each, which a graph does not remove)")
    return out


def sketch_workload_text(plan, proj, rows, features, dtype_name):
    """Where S lives for the call a fewbit_hip_sketch_describe plan belongs to: `s_fragment_bytes` == 0 <=> S is generated in the
    registers of the product kernel and never materialised; otherwise it is written once per call into the workspace, as the
    MFMA A fragments the product kernel reads back (tests/test_gpu_bench.py pins this equivalence)."""
    frag = int(plan.get('s_fragment_bytes', 0))
    where = ('generated in registers (never materialised)' if frag == 0 else
             f'written once per call into the workspace as MFMA A fragments ({frag} B) and read back by the product kernel')
    return f'out = S . M, S {proj} x {rows} {where}, M {rows} x {features} {dtype_name}'


SKETCH_SETTLE_S = 0.04         # as the headline's steady figures: the same launches for >= 40 ms first (clock transient after idle)
SKETCH_REPS = 100              # 3 x 100 timed launches per figure (tools/profile_sketch.sh: >= 200 dispatches under rocprofv3)


def settled_us(f, reps=SKETCH_REPS, rounds=3):
    """median over `rounds` of the average of `reps` back-to-back launches between two HIP events, after >= 40 ms of the same launches"""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < SKETCH_SETTLE_S:
        for _ in range(10):
            f()
        torch.cuda.synchronize()
    return sorted(event_time_us([f], reps, preroll=2) for _ in range(rounds))[rounds // 2]


def dct_intermediate_bytes(rows, features):
    """the fp32 intermediate of the sampled-DCT kernel pair: 64-feature tiles of rows x 32 complex (the rest of its workspace, the sorted
    samples, is 8 bytes per sample)"""
    return ((features + 63) // 64) * rows * 256


def measure_sampled_transform(device, rows, features, proj, dtype, dense_rademacher_us=None):
    """The reference's O(n log n) estimators (fewbit/functional/linear.py:113-131: `dct(input_view, dim=0, norm='ortho')[proj]`
    and `T.fft.fft(...)[proj]`) as this package runs them on the GPU (fewbit_amd.linear._sketch), beside their byte floor -- read M
    once, write the p sampled rows -- and the dense Rademacher sketch of the same shape."""
    from fewbit_amd import linear
    m = torch.randn(rows, features, device=device).to(dtype)
    gen = torch.Generator(device=device).manual_seed(3)
    es = m.element_size()
    floor_bytes = rows * features * es + proj * features * es
    rec = {'workload': f"dct(M, dim=0, norm='ortho')[idx], M {rows} x {features} {str(dtype).split('.')[-1]}, {proj} sampled rows",
           'byte_floor': {'bytes': floor_bytes, 'us_at_8TBs': round(floor_bytes / HBM_PEAK_GBS / 1e3, 2)}}
    for kind in ('dct', 'dft'):
        us = settled_us(lambda: linear.sampled_transform(kind, m, proj, gen), reps=20)
        rec[kind] = {'us': round(us, 1), 'x_byte_floor': round(us / (floor_bytes / HBM_PEAK_GBS / 1e3), 1), 'path': linear.sampled_transform_path(kind, m)}
        if kind == 'dct' and 'fewbit_hip_sampled_dct' in rec[kind]['path']:
            # the kernel pair's own roofline: the bytes its design moves (M once, the fp32 intermediate out and back, the sampled rows)
            moved = rows * features * es + 2 * dct_intermediate_bytes(rows, features) + proj * features * es
            rec[kind]['roofline'] = {'bound': 'hbm', 'bytes_moved_by_design': moved, 'achieved': round(moved / us / 1e3, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                     'frac': round(moved / us / 1e3 / HBM_PEAK_GBS, 4)}
    if dense_rademacher_us is not None:
        rec['dense_rademacher_us'] = dense_rademacher_us
    return rec


def measure_sketch(device):
    """the random-projection product of the randomized linear layers (fewbit_hip_sketch) where the reference quotes
    it: proj_dim_ratio 0.2 of 16384 tokens (README "Randomized Linear (20 %)", p = 3276) at both layer widths of RoBERTa-base
    (3072 and 768 features), bf16, both distributions, each with its roofline (class MFMA: 2*p*rows*features flops against the
    2.5 PFLOP/s dense bf16 peak) and beside what it replaces (S drawn into HBM + torch.matmul, timed in the same process).
    `wins_vs_torch` makes a regression visible in the line.  Every figure is settled (settled_us).  p = 1638 (ratio 0.1) stays as
    an extra.  `sampled_transform`: the reference's 'dct' / 'dft' estimators at the same shapes."""
    from fewbit_amd import cabi
    rows = 16384

    def one(features, proj, fp32_too):
        m = torch.randn(rows, features, device=device).to(torch.bfloat16)
        flops = 2.0 * proj * rows * features
        rec = {'flops': flops}
        ws = torch.empty(max(max(cabi.sketch_workspace_bytes(d, rows, features, proj) for d in cabi.SKETCH_DISTS), 1), dtype=torch.uint8, device=device)
        o = torch.empty(proj, features, dtype=torch.bfloat16, device=device)
        S = torch.randn(proj, rows, device=device, dtype=torch.bfloat16)
        rec['torch'] = {'randn_plus_matmul_us': round(settled_us(lambda: torch.randn(proj, rows, device=device, dtype=torch.bfloat16) @ m), 1),
                        'randint_plus_matmul_us': round(settled_us(lambda: (torch.randint(0, 2, (proj, rows), device=device, dtype=torch.int8).to(torch.bfloat16) * 2 - 1) @ m), 1),
                        'matmul_alone_us': round(settled_us(lambda: S @ m), 1)}
        del S
        for dist, pair in (('rademacher', 'randint_plus_matmul_us'), ('gaussian', 'randn_plus_matmul_us')):
            us = settled_us(lambda: cabi.sketch(dist, m, proj, 1234, 1.0 / proj, out=o, workspace=ws))
            plan = cabi.describe_sketch(dist, rows, features, proj)
            rec[dist] = {'workload': sketch_workload_text(plan, proj, rows, features, 'bf16'), 'us': round(us, 1), 'torch_us': rec['torch'][pair],
                         'wins_vs_torch': bool(us <= rec['torch'][pair]), 'plan': plan,
                         's_fragment_bytes': plan['s_fragment_bytes'], 'workspace_bytes': plan['workspace_bytes'],
                         'roofline': {'bound': 'mfma', 'achieved': round(flops / us / 1e6, 1), 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(flops / us / 1e6 / 2500.0, 4),
                                      'note': 'flops of the product only over the time of the whole call (fragment launch, product, slice reduction)'}}
        if fp32_too:    # fp32 input (the reference's own dtype): bf16 OPERANDS -- M is rounded to bf16 (once, into the workspace), fp32 sums and result
            m32 = m.float()
            ws32 = torch.empty(max(max(cabi.sketch_workspace_bytes(d, rows, features, proj, torch.float32) for d in cabi.SKETCH_DISTS), 1), dtype=torch.uint8, device=device)
            o32 = torch.empty(proj, features, dtype=torch.float32, device=device)
            rec['fp32_input'] = {}
            for dist in ('rademacher', 'gaussian'):
                plan = cabi.describe_sketch(dist, rows, features, proj, torch.float32)
                rec['fp32_input'][dist] = {'workload': sketch_workload_text(plan, proj, rows, features, 'fp32 (bf16 operands)'),
                                           'us': round(settled_us(lambda: cabi.sketch(dist, m32, proj, 1234, 1.0 / proj, out=o32, workspace=ws32)), 1),
                                           'operands': 'bf16 operands: the fp32 input is rounded to bf16 on its way into the matrix pipe, S likewise; fp32 accumulation and result '
                                                       '(the reference multiplies fp32 randn by fp32 input)',
                                           'converted_to_bf16_first': plan['converted_to_bf16_first'], 's_fragment_bytes': plan['s_fragment_bytes']}
        return rec

    out = {'note': 'proj_dim_ratio 0.2 (p = 3276 of 16384 rows) is the ratio of the reference README and tools/roberta_bench.py; 0.1 is an extra; '
                   f'every figure: >= {int(SKETCH_SETTLE_S * 1e3)} ms of the same launches first, then the median of 3 x {SKETCH_REPS} launches between HIP events',
           'ratio_0.2': {'16384x3072': one(3072, 3276, True), '16384x768': one(768, 3276, True)},
           'ratio_0.1': {'16384x3072': one(3072, 1638, False)}}
    out['wins_vs_torch'] = all(out['ratio_0.2'][shape][dist]['wins_vs_torch'] for shape in out['ratio_0.2'] for dist in ('rademacher', 'gaussian'))
    out['sampled_transform'] = {f'16384x{features}_{name}': _guarded(f'sampled_transform.{features}.{name}', lambda features=features, dt=dt: measure_sampled_transform(
                                    device, rows, features, 3276, dt, out['ratio_0.2'][f'16384x{features}']['rademacher']['us'] if dt == torch.bfloat16 else
                                    out['ratio_0.2'][f'16384x{features}'].get('fp32_input', {}).get('rademacher', {}).get('us')))
                                for features in (768, 3072) for name, dt in (('bf16', torch.bfloat16), ('fp32', torch.float32))}
    out['evidence'] = 'profiles/r06_sketch_bench.json, profiles/r06_sketch_rocprof_*_p3276_bf16.txt (tools/profile_sketch.sh), DESIGN.md section 7.3'
    return out


def cpu_baseline(name, cfg, reps, threads=None):
    """Reference CPU path on this host (oracle/_ref, kind 'reference': the reference's quantize/quantize_backward for the
    table configs, its bit codec Deflate/Inflate(..., 1) for the 1-bit config, exactly what BASELINE.md section 2 timed) or,
    without the prebuilt reference, the C restatement (kind 'port').  Bounded: `reps` repetitions of the full tensor."""
    tables = str(ROOT / 'fewbit_amd' / 'data' / 'builtin.npz')
    ref_dir = ROOT / 'oracle' / '_ref'
    rows, cols, bits, dname = cfg['rows'], cfg['cols'], cfg['bits'], cfg['dtype']
    nbytes, _ = step_bytes(cfg)
    have_ref = (ref_dir / ('libcodec_ref.so' if cfg['kind'] == 'stepwise1' else 'libfewbit_ref.so')).exists()
    if have_ref:
        try:
            cmd = [sys.executable, str(ROOT / 'oracle' / 'ref_bench.py'), str(rows), str(cols), dname, str(bits), str(reps), tables,
                   str(threads or 0), cfg['fn']]
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, check=True)
            r = json.loads(out.stdout.strip().splitlines()[-1])
            return {'value': round(r['gib_per_s'], 4), 'unit': 'GiB/s', 'cores': r['threads'], 'kind': 'reference',
                    'host_cpus': r['cores'], 'ms_per_step': round(r['seconds_per_step'] * 1e3, 2),
                    'sample': f'{reps} x ({r["what"]}) of the full {rows}x{cols} {dname} tensor, median; {r["threads"]} thread(s)'}
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f'[bench] reference CPU baseline for {name} failed ({e}); falling back to the C port\n')
    import oracle  # the checker, timed as the CPU baseline only
    dt = DTYPES[dname]
    torch.manual_seed(0)
    x = torch.randn(rows, cols).to(dt)
    torch.manual_seed(1)
    gy = torch.randn(rows, cols).to(dt)
    if cfg['kind'] == 'continuous':
        with np.load(tables) as z:
            borders = torch.tensor(z[f"{cfg['fn']}{bits:02d}-borders"]).to(dt)[1:-1].contiguous()
            levels = torch.tensor(z[f"{cfg['fn']}{bits:02d}-levels"]).to(dt)
    times = []
    for i in range(min(reps, 3) + 1):
        t0 = time.perf_counter()
        if cfg['kind'] == 'continuous':
            _, state, _ = oracle.quantize(cfg['fn'], x, borders)
            oracle.quantize_backward(gy, state, levels)
        else:
            _, state = oracle.stepwise1_forward(cfg['fn'], x)
            oracle.stepwise1_backward(cfg['fn'], gy, state)
        t1 = time.perf_counter()
        if i:
            times.append(t1 - t0)
    best = float(np.median(times))
    return {'value': round(nbytes / best / 2**30, 4), 'unit': 'GiB/s', 'cores': 1, 'kind': 'port',
            'sample': f'{len(times)} x (forward + backward) of the full {rows}x{cols} {dname} tensor, median; '
                      'oracle/fewbit_oracle.c, scalar, 1 thread', 'ms_per_step': round(best * 1e3, 2)}


# ---- synchronisation of the ranks around the timed region (control plane only; the data path has no exchange) ----
class NoSync:
    def barrier(self, tag):
        pass

    def max(self, values):
        return values

    def close(self):
        pass


class FileSync:
    """Barrier through files in a directory the parent created (self-launched N > 1): rank r creates `<tag>.<r>` and spins
    until all `world` files exist.  No sockets, no RCCL."""

    def __init__(self, directory, rank, world, timeout=300.0):
        self.dir, self.rank, self.world, self.timeout = Path(directory), rank, world, timeout

    def barrier(self, tag):
        (self.dir / f'{tag}.{self.rank}').touch()
        names = [self.dir / f'{tag}.{r}' for r in range(self.world)]
        t0 = time.perf_counter()
        while not all(p.exists() for p in names):
            if (self.dir / 'abort').exists() or time.perf_counter() - t0 > self.timeout:
                raise RuntimeError(f'rank {self.rank}: barrier `{tag}` failed (abort flag or timeout)')
            time.sleep(0.0005)      # outside the timed region; N ranks x N files must not burn N cores while the slowest sets up

    def max(self, values):          # the parent takes the max over the ranks' result files
        return values

    def close(self):
        pass


class TorchSync:
    """Under torch.distributed.run: barrier + max over ranks through a process group -- gloo on the CPU by default
    (FEWBIT_BENCH_BACKEND=nccl: RCCL, for the same two calls)."""

    def __init__(self, rank, world, device):
        import torch.distributed as dist
        self.dist, self.device = dist, device
        self.backend = os.environ.get('FEWBIT_BENCH_BACKEND', 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if os.environ['MASTER_ADDR'] in ('127.0.0.1', 'localhost', '::1'):
            # one node: the loopback interface, whatever the container's hostname resolves to (or fails to)
            os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
        if self.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(self.backend, rank=rank, world_size=world)

    def barrier(self, tag):
        self.dist.barrier()

    def max(self, values):
        t = torch.tensor(values, dtype=torch.float64, device=self.device if self.backend == 'nccl' else 'cpu')
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    def close(self):
        self.dist.destroy_process_group()


def run_rank(args, rank, local_rank, world, make_sync):
    """The timed region on one rank.  Returns this rank's measurements (after sync.max: the maxima over ranks)."""
    ndev = max(torch.cuda.device_count(), 1)
    device = torch.device('cuda', local_rank % ndev)
    torch.cuda.set_device(device)
    sync = make_sync(device)
    from fewbit_amd import cabi   # raises if libfewbit_hip.so is missing: there is no fallback
    from fewbit_amd.sharding import shard_range
    cabi.lib()

    cfg = CONFIGS[args.config]
    n = cfg['rows'] * cfg['cols']
    if args.scaling == 'strong':
        # ONE tensor of n elements (seed 0), rank r owns -- and only ever uploads -- elements [begin, end) of it
        # (--emulate-world M on one GPU: rank 0's slice of an M-way cut, alone on the device -- what each of M separate
        # GPUs would run, since nothing is shared between ranks)
        begin, end = shard_range(n, args.emulate_world or world, rank)
        w = Workload(cfg, device, nsets=1, seed=0, span=(begin, end))
    else:
        # which elements of the (weak-scaled) global tensor this rank owns: documentation of the cut, the data is synthetic
        begin, end = shard_range(n * world, world, rank)
        assert end - begin == n
        w = Workload(cfg, device, nsets=1, seed=rank)
    fwd, bwd = w.fwd[0], w.bwd[0]

    t_warm = time.perf_counter()
    for _ in range(args.warmup):
        fwd()
        bwd()
    settle_steps = 0
    if args.settle_ms > 0:      # optional: a GPU that was idle boosts ~1.5 ms, dips ~10 ms, then settles (scratch/timeline.py)
        torch.cuda.synchronize()
        while (time.perf_counter() - t_warm) * 1e3 < args.settle_ms:
            for _ in range(50):
                fwd()
                bwd()
            settle_steps += 50
            torch.cuda.synchronize()
    # ---- the timed region: barrier + synchronize | host clock | EXACTLY K steps | synchronize | host clock | barrier.
    # Nothing else inside: no collective, no event, no host synchronisation (an event record costs ~5 us of a short
    # region, spinning on an event query instead of the blocking synchronize costs MORE: scratch/wall_overhead.py,
    # profiles/r03_wall_overhead.txt).
    torch.cuda.synchronize()
    sync.barrier('start')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fwd()
        bwd()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    sync.barrier('stop')
    # the same K steps once more between two HIP events on the launch stream: the GPU-side time of such a region (without
    # the first launch's latency from an idle queue and the wake-up after the synchronize, ~18 us together)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    e1.record()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.steps):
        fwd()
        bwd()
    e1.record()
    torch.cuda.synchronize()
    event = e0.elapsed_time(e1) * 1e-3
    wall_max, event_max = sync.max([wall, event])
    res = {'rank': rank, 'device': device.index, 'n_devices': ndev, 'wall_s': wall_max, 'event_s': event_max,
           'own_wall_s': wall, 'own_event_s': event, 'settle_steps': settle_steps, 'set_bytes': w.set_bytes,
           'elements': w.n, 'span': [begin, end]}
    if args.digests:
        res['sha256'] = w.digests()
    if rank == 0:
        # Per-kernel figures for `roofline`, measured AFTER the contract's region and settled: a GPU that was idle boosts for
        # ~1.5 ms, then runs 8-12 % slower for ~10 ms, then settles (scratch/timeline.py), and a short timed region (the
        # driver's K = 20 is 0.5 ms) sits inside that transient.  ~40 ms of the same steps first, then: a long alternating
        # fwd/bwd region between two events (steady_step_us), and each kernel back to back (fwd_us, bwd_us).
        t_s = time.perf_counter()
        while (time.perf_counter() - t_s) < 0.04:
            for _ in range(50):
                fwd()
                bwd()
            torch.cuda.synchronize()
        reps = max(400, min(args.steps, 2000))
        res['steady_step_us'] = event_time_us([fwd, bwd], reps, preroll=4)
        res['fwd_us'], res['bwd_us'] = event_time_us([fwd], reps), event_time_us([bwd], reps)
        res['kernels'] = describe(cfg, w.n)
    return res, sync, device, w


METRICS = {
    'c2': 'fwd+bwd GiB/s (and % HBM roofline) for 3-bit GELU, 4096x4096 bf16',
    'c4': 'fwd+bwd GiB/s (and % HBM roofline) for 3-bit GELU, 4x(16384x4096) bf16 sharded over 8 GPUs (per-GPU shard 8192x4096)',
    'c4_tensor': 'fwd+bwd GiB/s (and % HBM roofline) for 3-bit GELU, one 16384x4096 bf16 tensor of 4x(16384x4096)',
}


def check_placement(world, n_devices, devices):
    """With at least as many devices as ranks every rank must have landed on its own device."""
    if devices is None or world > n_devices:
        return
    if len(set(devices)) != len(devices):
        raise RuntimeError(f'{world} ranks on a box with {n_devices} devices, but they did not land on pairwise distinct devices: {devices}')


def build_line(args, world, res, per_rank=None, launcher='single process'):
    cfg = CONFIGS[args.config]
    n = cfg['rows'] * cfg['cols']
    strong = args.scaling == 'strong'
    n_rank0 = res.get('elements', n)                     # rank 0 owns the largest slice (shard_range gives extras to low ranks)
    sb, _ = step_bytes(cfg)                              # whole config tensor
    _, fb = step_bytes(cfg, n_rank0)                     # what ONE forward launch of rank 0 moves
    wall, event = res['wall_s'], res['event_s']
    total = sb * args.steps * (1 if strong else world)
    value = total / wall / 2**30
    event_step_us = event / args.steps * 1e6
    fwd_us, bwd_us = res['fwd_us'], res['bwd_us']
    # forward's duration inside a step: the GPU time of a step (HIP events on the launch stream around a long, settled
    # fwd/bwd region -- see run_rank) split in the ratio of the two kernels' stand-alone durations; rocprofv3's per-dispatch
    # average of the same command (profiles/) is the cross-check.  The same split of the contract's own K-step region is
    # given beside it (frac_timed_region): with a short K it sits in the clock transient after idle.
    steady = res['steady_step_us']
    fwd_in_step_us = steady * fwd_us / (fwd_us + bwd_us)
    achieved = fb / (fwd_in_step_us * 1e-6) / 1e9
    fwd_in_timed_us = event_step_us * fwd_us / (fwd_us + bwd_us)
    traffic, traffic_source = None, None
    tf = ROOT / 'profiles' / 'traffic_forward.json'
    if args.config == 'c2' and world == 1 and tf.exists():
        try:
            doc = json.loads(tf.read_text())
            traffic = doc.get('hbm_bytes_per_launch')
            traffic_source = (f"replayed from {doc.get('source', 'profiles/traffic_forward.json')}: rocprofv3 --pmc FETCH_SIZE / "
                              "WRITE_SIZE passes over `bench.py --no-extras` (tools/profile_round.sh), NOT measured in this run")
        except Exception:  # noqa: BLE001
            traffic = None
    kf, kb = res['kernels']
    shared = world > res['n_devices']
    devices_used = min(world, res['n_devices'])          # ranks sharing a GPU share its 8 TB/s too
    if per_rank:
        check_placement(world, res['n_devices'], [r['device'] for r in per_rank])
    if strong:
        what = (f"ONE {cfg['rows']}x{cfg['cols']} {cfg['dtype']} tensor ({cfg['fn']} bits={cfg['bits']}) cut into {world} slice(s) by "
                "sharding.shard_range, one slice per GPU")
        parallelism = f'{world} slice(s) of one {n}-element tensor (largest: {n_rank0} elements), cut by sharding.shard_range, no collectives'
    else:
        what = f"{cfg['label']} per GPU"
        parallelism = f'{world} independent shard(s) of {n} elements, cut by sharding.shard_range, no collectives'
    line = {
        'metric': METRICS[args.config],
        'value': round(value, 2), 'unit': 'GiB/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(wall / args.steps * 1e3, 5), 'higher_is_better': True, 'scaling': args.scaling,
        'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': f"{what}, fused quantize+pack fwd / unpack+mul bwd via the C-ABI, inputs resident in HBM",
                   'name': args.config, 'elements_per_gpu': n_rank0, 'elements_total': n if strong else n * world,
                   'bytes_per_step_per_gpu': int(step_bytes(cfg, n_rank0)[0]), 'bytes_per_step_total': int(sb * (1 if strong else world)),
                   'parallelism': parallelism,
                   'launcher': launcher,
                   'working_set_MiB_per_gpu': round(res['set_bytes'] / 2**20, 1),
                   'cache_state': 'warm: one buffer set re-used every step; it fits the 256 MiB Infinity Cache (see `cold`)'
                                  if res['set_bytes'] < INFINITY_CACHE_BYTES else 'one buffer set, larger than the 256 MiB Infinity Cache'},
        'timing': {'method': 'host wall clock around exactly K steps, bracketed by barrier + synchronize on both sides; max over ranks '
                             '(value, ms_per_step)',
                   'event_ms_per_step': round(event / args.steps * 1e3, 5),
                   'event_GiB_s': round(total / event / 2**30, 2),
                   'event_note': 'a second, identical K-step region right after the timed one, between two HIP events on the launch '
                                 'stream: GPU-side time of such a region, without the first launch\'s latency from an idle queue and the '
                                 'wake-up after the final synchronize (~18 us together, profiles/r03_wall_overhead.txt)',
                   'warmup_settle': (f'{res["settle_steps"]} extra untimed steps until the GPU had been busy {args.settle_ms:g} ms'
                                     if args.settle_ms > 0 else 'off: exactly W warm-up steps')},
        'pct_of_hbm_roofline': round(100.0 * (total / wall / 1e9) / (HBM_PEAK_GBS * devices_used), 2),
        'pct_of_hbm_roofline_event_timed': round(100.0 * (total / event / 1e9) / (HBM_PEAK_GBS * devices_used), 2),
        'fwd_us': round(fwd_us, 2), 'bwd_us': round(bwd_us, 2), 'fwd_in_step_us': round(fwd_in_step_us, 2),
        'roofline': {'bound': 'hbm', 'kernel': kf['kernel'], 'launch_shape': {k: kf[k] for k in ('blocks', 'threads', 'blocks_per_cu', 'chunk', 'u')},
                     'other_kernel': kb['kernel'],
                     'achieved': round(achieved, 1),
                     'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                     'traffic': traffic, 'traffic_source': traffic_source, 'algorithmic_bytes_per_launch': int(fb),
                     'avg_launch_us': round(fwd_in_step_us, 2),
                     'avg_launch_us_method': 'GPU time per step of a settled >= 400-step fwd/bwd region (two HIP events on the launch stream, '
                                             'after the contract\'s region and ~40 ms of the same steps) x fwd/(fwd+bwd) of the back-to-back '
                                             'per-kernel HIP-event timings (fwd_us, bwd_us)',
                     'steady_step_us': round(steady, 2),
                     'frac_timed_region': round(fb / (fwd_in_timed_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     'frac_timed_region_note': f'the same split applied to a {args.steps}-step region right after the contract\'s own (event-timed, timing.event_ms_per_step)',
                     'cache_state': 'warm (x and gy are served from the Infinity Cache; writes go to HBM); frac_cold is the figure '
                                    'with nothing cached'},
    }
    if strong and args.emulate_world:
        m = args.emulate_world
        line['emulated_world'] = m
        line['projected'] = {'n_gpus': m, 'value_GiB_s': round(sb * args.steps / wall / 2**30, 2),
                             'pct_of_hbm_roofline': round(100.0 * (sb * args.steps / wall / 1e9) / (HBM_PEAK_GBS * m), 2),
                             'note': f'this GPU ran rank 0\'s slice ({n_rank0} of {n} elements) of a {m}-way cut ALONE; the path has no exchange step, so {m} '
                                     f'separate GPUs each take this time for their slice: projected whole-tensor rate = {int(sb)} B / this time. '
                                     'A projection from one device, not a measurement on several.'}
        line['value'] = round(step_bytes(cfg, n_rank0)[0] * args.steps / wall / 2**30, 2)     # what THIS device did
        line['pct_of_hbm_roofline'] = round(100.0 * (step_bytes(cfg, n_rank0)[0] * args.steps / wall / 1e9) / HBM_PEAK_GBS, 2)
        line['config']['bytes_per_step_total'] = int(step_bytes(cfg, n_rank0)[0])
    if strong and (world > 1 or args.emulate_world):
        line['roofline']['note'] = (f'rank 0\'s slice ({n_rank0} elements, {int(fb)} B per forward launch): at this size a launch is bounded by the '
                                    'dispatch boundary (an empty kernel is 2.5-2.7 us), not by HBM')
    if shared:
        line['shared_gpu'] = True
        line['config']['parallelism'] += (f' -- {world} ranks on {res["n_devices"]} visible device(s): ranks SHARE a GPU (validation run; '
                                          f'pct_of_hbm_roofline is against {devices_used} x 8 TB/s)')
    if per_rank:
        line['per_gpu_us_per_step'] = [round(r['own_wall_s'] / args.steps * 1e6, 2) for r in per_rank]
        line['per_gpu_event_us_per_step'] = [round(r['own_event_s'] / args.steps * 1e6, 2) for r in per_rank]
        line['per_gpu_device'] = [r['device'] for r in per_rank]
        if all('elements' in r for r in per_rank):
            line['per_gpu_elements'] = [r['elements'] for r in per_rank]
        if all('sha256' in r for r in per_rank):
            line['per_gpu_sha256'] = [r['sha256'] for r in per_rank]
            line['per_gpu_span'] = [r['span'] for r in per_rank]
    elif 'sha256' in res:
        line['per_gpu_sha256'], line['per_gpu_span'] = [res['sha256']], [res['span']]
    line['evidence'] = {'rocprofv3_kernel_stats': 'profiles/r04_bench_kernel_stats.csv (this command with --no-extras)',
                        'pmc_traffic': 'profiles/r04_pmc_traffic.json', 'per_config_rocprofv3_and_pmc': 'profiles/r04_configs.json',
                        'shape_sweeps': 'profiles/r03_shape_sweep_*.txt, profiles/r04_state_staging_ab.txt',
                        'copy_floor_at_this_size': 'profiles/r04_stream_bench_32MiB.txt',
                        'multi_rank_launch_paths': 'profiles/r04_bench_line_{4,8}ranks_*_shared.json, profiles/r04_strong_*',
                        'regenerate': 'bash tools/profile_round.sh r04'}
    return line


def under_profiler():
    e = os.environ
    return any(k.startswith(('ROCPROF', 'ROCPROFILER_', 'ROCP_')) for k in e) or 'rocprofiler' in e.get('LD_PRELOAD', '')


def measure_traffic(args, algorithmic, steps=50):
    """HBM bytes per launch of the forward (dominant) kernel from the PMC counters, measured by THIS run: two child processes,
    `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` (separate passes, nothing else traced), each
    over `bench.py --no-extras --no-cpu-baseline` of the same config; per-dispatch counter rows of the forward kernel averaged;
    gfx950 correction of MI355X_MICROARCH.md's HBM section: FETCH_SIZE (KiB) counts half of a wide coalesced read -> doubled,
    WRITE_SIZE (KiB) as is.  The same passes run from tools/profile_round.sh are kept under profiles/ as the cross-check."""
    import csv
    tool = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(tool):
        raise RuntimeError('rocprofv3 not found')
    if under_profiler():
        raise RuntimeError('already running under a profiler')
    means, counts = {}, {}
    with tempfile.TemporaryDirectory(prefix='fewbit_pmc_', dir='/tmp') as tmp:
        for name in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(tmp, name)
            cmd = [tool, '--pmc', name, '--kernel-trace', '--output-format', 'csv', '-d', out, '-o', 'pmc', '--',
                   sys.executable, str(Path(__file__).resolve()), '--config', args.config, '--steps', str(steps), '--warmup', '5',
                   '--no-extras', '--no-cpu-baseline']
            env = dict(os.environ, TMPDIR='/tmp')
            run = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=240)
            if run.returncode != 0:
                raise RuntimeError(f'rocprofv3 --pmc {name} rc={run.returncode}: {run.stderr[-200:]}')
            vals = []
            for f in Path(out).rglob('*counter_collection.csv'):
                with open(f, newline='') as fh:
                    for row in csv.DictReader(fh):
                        if row['Counter_Name'] == name and 'fewbit_hip::' in row['Kernel_Name'] and 'forward' in row['Kernel_Name']:
                            vals.append(float(row['Counter_Value']))
            if not vals:
                raise RuntimeError(f'no {name} rows for the forward kernel')
            means[name], counts[name] = sum(vals) / len(vals), len(vals)
    fetch, write = int(round(means['FETCH_SIZE'] * 1024 * 2)), int(round(means['WRITE_SIZE'] * 1024))
    return {'hbm_bytes_per_launch': fetch + write, 'fetch_bytes_corrected': fetch, 'write_bytes': write,
            'FETCH_SIZE_KiB_raw': round(means['FETCH_SIZE'], 1), 'WRITE_SIZE_KiB': round(means['WRITE_SIZE'], 1),
            'dispatches': counts['FETCH_SIZE'], 'over_algorithmic': round((fetch + write) / algorithmic, 4),
            'source': 'measured by this run: rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (two child '
                      f'processes) over `bench.py --config {args.config} --steps {steps} --warmup 5 --no-extras --no-cpu-baseline`; FETCH_SIZE x2 '
                      '(gfx950), KiB units; per launch of the forward kernel'}


def _guarded(what, fn):
    """The extras never take the contract's line down with them: a failure is recorded in place of the result."""
    try:
        return fn()
    except Exception as e:  # noqa: BLE001
        sys.stderr.write(f'[bench] extra `{what}` failed: {type(e).__name__}: {e}\n')
        torch.cuda.empty_cache()
        return {'error': f'{type(e).__name__}: {e}'[:300]}


def _other_config(name, c, args, device):
    kf, kb = describe(c)
    entry = {'workload': c['label'], 'bytes_per_step': int(step_bytes(c)[0]),
             'kernels': {'fwd': kf['kernel'], 'bwd': kb['kernel'],
                         'fwd_shape': {k: kf[k] for k in ('blocks', 'threads', 'chunk')},
                         'bwd_shape': {k: kb[k] for k in ('blocks', 'threads', 'chunk')}},
             'warm': measure_config(c, device, cold=False)}
    if step_bytes(c)[0] < (64 << 20):                    # launch-bound size: add the host-free figure
        def graph():
            us = graph_step_us(c, device)
            return {'us_step': round(us, 2), 'GiB_s': round(step_bytes(c)[0] / (us * 1e-6) / 2**30, 1),
                    'frac': round(step_bytes(c)[0] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                    'note': '50 steps captured in one hipGraph, replayed: kernels + their two dependent-launch boundaries, no host '
                            'launch cost (the eager figures above are bounded by the host: two launches per step at >= 2.7 us each)'}
        entry['warm_hipgraph'] = _guarded(f'{name}.warm_hipgraph', graph)
    if step_bytes(c)[0] < 8 * INFINITY_CACHE_BYTES:      # beyond that one buffer set is cache-cold by itself
        entry['cold'] = measure_config(c, device, cold=True)
    if 'reference_published' in c:
        entry['reference_published'] = c['reference_published']
        entry['vs_reference_published'] = round(entry['warm']['GiB_s'] / c['reference_published']['GiB_s'], 2)
    if not args.no_cpu_baseline and name in CPU_SAMPLES:
        entry['cpu_baseline'] = _guarded(f'{name}.cpu_baseline', lambda: cpu_baseline(name, c, CPU_SAMPLES[name][0]))
    return entry


def add_extras(line, args, device):
    cfg = CONFIGS[args.config]
    _, fb = step_bytes(cfg)
    if not args.no_extras:
        cold = _guarded('cold', lambda: measure_config(cfg, device, cold=True))
        line['cold'] = cold
        if 'error' not in cold:
            cold['note'] = ('same workload, rotating through independent buffer sets so nothing is re-used from L2 / Infinity Cache; '
                            'frac = fwd+bwd algorithmic bytes / us_step / 8 TB/s')
            line['roofline']['frac_cold'] = round(fb / (cold['us_fwd'] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            line['roofline']['avg_launch_us_cold'] = cold['us_fwd']
        line['op_level'] = _guarded('op_level', lambda: measure_op_level(cfg, device))
        line['configs'] = {name: _guarded(f'configs.{name}', lambda name=name, c=c: _other_config(name, c, args, device))
                           for name, c in CONFIGS.items() if name not in (args.config, 'c4_tensor')}
        line['sketch'] = _guarded('sketch', lambda: measure_sketch(device))
        if not args.no_pmc:
            live = _guarded('pmc_traffic', lambda: measure_traffic(args, fb))
            line['roofline']['traffic_live'] = live
            if 'error' not in live:                      # this run's own counters replace the recorded figure
                line['roofline']['traffic_recorded'] = line['roofline']['traffic']
                line['roofline']['traffic'], line['roofline']['traffic_source'] = live['hbm_bytes_per_launch'], live['source']
                line['roofline']['traffic_live_error'] = None
            else:                                        # visible in the line, not only on stderr: `traffic` is then the recorded figure
                line['roofline']['traffic_live_error'] = live['error']
    if not args.no_cpu_baseline:
        reps_all, reps_one = CPU_SAMPLES[args.config]
        line['cpu_baseline'] = _guarded('cpu_baseline', lambda: cpu_baseline(args.config, cfg, reps_all))
        if reps_one:
            line['cpu_baseline_1thread'] = _guarded('cpu_baseline_1thread', lambda: cpu_baseline(args.config, cfg, reps_one, threads=1))


def parent_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start N children (one per device) BEFORE this process touches
    the GPU, wait for them, print the one line.  Never exec: the children are ordinary subprocesses."""
    world = args.gpus
    sync_dir = Path(tempfile.mkdtemp(prefix='fewbit_bench_'))
    children = []
    try:
        # FEWBIT_BENCH_WORKER: a stand-in worker script speaking the same protocol (tests/test_bench_launcher.py, no GPU)
        worker = os.environ.get('FEWBIT_BENCH_WORKER') or str(Path(__file__).resolve())
        base = [sys.executable, worker, '--gpus', str(world), '--steps', str(args.steps), '--warmup', str(args.warmup),
                '--config', args.config, '--scaling', args.scaling, '--settle-ms', str(args.settle_ms), '--sync-dir', str(sync_dir)]
        if args.digests:
            base.append('--digests')
        env = dict(os.environ)
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
            env.pop(k, None)
        for r in range(world):
            children.append(subprocess.Popen(base + ['--worker-rank', str(r)], env=env, stdout=subprocess.DEVNULL))
        deadline = time.time() + args.launch_timeout
        failed = None
        pending = set(range(world))
        while pending and failed is None:
            for r in sorted(pending):
                rc = children[r].poll()
                if rc is not None:
                    pending.discard(r)
                    if rc != 0:
                        failed = (r, rc)
            if time.time() > deadline:
                failed = (-1, 'timeout')
            time.sleep(0.02)
        if failed is not None:
            (sync_dir / 'abort').touch()
            for c in children:
                if c.poll() is None:
                    c.terminate()            # exactly the PIDs started above
            for c in children:
                try:
                    c.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    c.kill()
            for ef in sorted(sync_dir.glob('error.*')):     # the REASON, not just the rank: each child's own traceback
                sys.stderr.write(f'---- bench.py rank {ef.suffix[1:]} raised ----\n{ef.read_text()}\n')
            sys.exit(f'bench.py: rank {failed[0]} failed ({failed[1]}); no result')
        per_rank = [json.loads((sync_dir / f'result.{r}.json').read_text()) for r in range(world)]
        res = dict(per_rank[0])
        res['wall_s'] = max(r['own_wall_s'] for r in per_rank)
        res['event_s'] = max(r['own_event_s'] for r in per_rank)
        try:
            line = build_line(args, world, res, per_rank,
                              launcher=f'bench.py started {world} child processes itself (one per device, file barriers, no process group)')
        except RuntimeError as e:                           # ranks did not land on distinct devices
            sys.exit(f'bench.py: {e}; no result')
        print(json.dumps(line), flush=True)
    finally:
        shutil.rmtree(sync_dir, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--config', choices=('c2', 'c4', 'c4_tensor'), default='c2', help='headline workload per GPU (see module docstring)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='weak (default, the driver\'s contract): every GPU gets its own tensor of the config\'s size; strong: ONE such tensor '
                         'is cut over the N GPUs by sharding.shard_range')
    ap.add_argument('--emulate-world', type=int, default=0,
                    help='with --gpus 1 --scaling strong: run rank 0\'s slice of an M-way cut alone on this GPU and add the projected '
                         'M-GPU figure to the line (a projection, labelled so)')
    ap.add_argument('--digests', action='store_true', help='add per-rank SHA-256 of (y, state, gx) to the line (after the timed region)')
    ap.add_argument('--settle-ms', type=float, default=0.0,
                    help='optional: keep issuing untimed warm-up steps until the GPU has been busy this long (default 0 = exactly W '
                         'steps): MI355X drops its clocks 1.5-10 ms after load begins and recovers by ~15 ms (scratch/timeline.py)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='timed region only (what the rocprofv3 passes run)')
    ap.add_argument('--no-pmc', action='store_true', help='skip the two rocprofv3 --pmc child runs behind roofline.traffic')
    ap.add_argument('--worker-rank', type=int, default=None, help=argparse.SUPPRESS)     # set by parent_launch
    ap.add_argument('--sync-dir', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--launch-timeout', type=float, default=900.0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.emulate_world and (args.gpus != 1 or args.scaling != 'strong' or 'WORLD_SIZE' in os.environ):
        ap.error('--emulate-world needs --gpus 1 --scaling strong and no launcher')
    if args.worker_rank is not None:                     # a child of parent_launch
        rank, world = args.worker_rank, args.gpus
        try:
            if os.environ.get('FEWBIT_BENCH_INJECT_FAILURE') == str(rank):      # test hook: a rank that raises (tests/test_gpu_bench.py)
                raise RuntimeError(f'injected failure on rank {rank}')
            res, sync, device, w = run_rank(args, rank, rank, world, lambda device: FileSync(args.sync_dir, rank, world))
            tmp = Path(args.sync_dir) / f'result.{rank}.json.tmp'
            tmp.write_text(json.dumps(res))
            tmp.rename(Path(args.sync_dir) / f'result.{rank}.json')
        except BaseException:                            # the parent prints this file: a child's stdout goes nowhere
            import traceback
            (Path(args.sync_dir) / f'error.{rank}').write_text(traceback.format_exc())
            raise
        return

    if 'WORLD_SIZE' in os.environ:                       # under torch.distributed.run
        rank, local_rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ['WORLD_SIZE'])
        args.gpus = world
        if world > 1:
            res, sync, device, w = run_rank(args, rank, local_rank, world, lambda device: TorchSync(rank, world, device))
            per_rank = None
            if sync.backend != 'nccl':                  # reporting only (after the timed region): each rank's own figures
                own = {k: res[k] for k in ('own_wall_s', 'own_event_s', 'device', 'elements', 'span', 'sha256') if k in res}
                per_rank = [None] * world
                sync.dist.all_gather_object(per_rank, own)
            if rank == 0:
                try:
                    line = build_line(args, world, res, per_rank,
                                      launcher=f'torch.distributed.run, {world} ranks; {sync.backend} process group for the barrier and the max only')
                except RuntimeError as e:                   # ranks did not land on distinct devices
                    sync.close()
                    sys.exit(f'bench.py: {e}; no result')
                print(json.dumps(line), flush=True)
            sync.close()
            return

    if args.gpus > 1:
        return parent_launch(args)

    res, sync, device, w = run_rank(args, 0, int(os.environ.get('LOCAL_RANK', 0)), 1, lambda device: NoSync())
    line = build_line(args, 1, res)
    del w
    torch.cuda.empty_cache()
    add_extras(line, args, device)
    print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
