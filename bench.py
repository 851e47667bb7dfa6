#!/usr/bin/env python3
"""bench.py -- forward+backward throughput of the few-bit activation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c4] [--no-extras] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config c4]

One "step" = one pass of the hot path over one batch: the fused quantize+pack forward and the fused unpack+mul
backward of fewbit.gelu(bits=3), both through the C-ABI (include/fewbit_hip.h) on HBM-resident synthetic tensors.
  --config c2 (default)  4096x4096 bf16 per GPU              (BASELINE.json configs[1], the headline metric)
  --config c4            8192x4096 bf16 per GPU = the shard one GPU owns of BASELINE.json configs[3]
                         (4 x (16384x4096) bf16 over 8 GPUs, cut by fewbit_amd.sharding.shard_range)
With N > 1 every rank runs the same per-GPU workload on its own shard (weak scaling, no collective on the data
path; torch.distributed carries only the barrier before and the max-over-ranks after the timed region).

Timed region: W warm-up steps, barrier + synchronize, a short untimed pre-roll that fills the launch queue,
then EXACTLY K steps between two HIP events recorded on the launch stream, synchronize (+ barrier).  ms_per_step is
the event time / K (max over ranks); the host wall clock around the same region is reported beside it.

metric = algorithmic bytes / time; algorithmic bytes per element = 4*s + k/4 (fwd: read x, write y, write state;
bwd: read gy, read state, write gx; s = element size, k = bits) -- SURVEY.md 8(d).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries (N = 1, unless --no-extras):
  cold       the same workload rotating through > 1 GiB of independent buffer sets (Infinity Cache out of the picture)
  configs    every other BASELINE config on one GPU (c1 relu fp32 1024^2, c3 silu k=2/k=4 fp16 8192^2, c4 shard,
             c2 in fp32), each cache-warm and cache-cold: us_fwd / us_bwd / us_step / GiB_s / frac of 8 TB/s
  cpu_baseline, cpu_baseline_1thread   the reference's own CPU path (oracle/_ref) on the host cores
"""
import argparse
import json
import os
import subprocess
import sys
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


def load_tables(cfg, device):
    dt = DTYPES[cfg['dtype']]
    with np.load(ROOT / 'fewbit_amd' / 'data' / 'builtin.npz') as z:
        key = f"{cfg['fn']}{cfg['bits']:02d}"
        borders = torch.tensor(z[f'{key}-borders']).to(dt)[1:-1].contiguous().to(device)
        levels = torch.tensor(z[f'{key}-levels']).to(dt).to(device)
    return borders, levels


def step_bytes(cfg):
    es = torch.empty(0, dtype=DTYPES[cfg['dtype']]).element_size()
    n = cfg['rows'] * cfg['cols']
    return n * (4 * es + cfg['bits'] / 4), n * (2 * es + cfg['bits'] / 8)


class Workload:
    """`nsets` independent buffer sets (x, y, gy, gx, state) of one config with pre-resolved launches per set."""

    def __init__(self, cfg, device, nsets=1, seed=0, host_seeded=True):
        from fewbit_amd import cabi
        dt = DTYPES[cfg['dtype']]
        n = cfg['rows'] * cfg['cols']
        self.cfg, self.n, self.nsets = cfg, n, nsets
        self.fwd, self.bwd, self.keep = [], [], []
        if cfg['kind'] == 'continuous':
            borders, levels = load_tables(cfg, device)
        for i in range(nsets):
            if host_seeded:     # synthetic shard: seeded on the host, then resident in HBM
                g = torch.Generator().manual_seed(2 * seed + 1000 * i)
                x = torch.randn(cfg['rows'], cfg['cols'], generator=g).to(dt).to(device)
                g = torch.Generator().manual_seed(2 * seed + 1 + 1000 * i)
                gy = torch.randn(cfg['rows'], cfg['cols'], generator=g).to(dt).to(device)
            else:               # the extra measurements: same distribution, drawn on the device
                g = torch.Generator(device=device).manual_seed(2 * seed + 1000 * i)
                x = torch.randn(cfg['rows'], cfg['cols'], generator=g, device=device).to(dt)
                gy = torch.randn(cfg['rows'], cfg['cols'], generator=g, device=device).to(dt)
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


def cpu_baseline(cfg, threads=None):
    """Reference CPU path (oracle/_ref, kind 'reference') or, without it, the C restatement (kind 'port')."""
    tables = str(ROOT / 'fewbit_amd' / 'data' / 'builtin.npz')
    ref = ROOT / 'oracle' / '_ref' / 'libfewbit_ref.so'
    rows, cols, bits, dname = cfg['rows'], cfg['cols'], cfg['bits'], cfg['dtype']
    nbytes, _ = step_bytes(cfg)
    if ref.exists():
        reps = 8 if threads == 1 else 20
        try:
            cmd = [sys.executable, str(ROOT / 'oracle' / 'ref_bench.py'), str(rows), str(cols), dname, str(bits), str(reps), tables]
            if threads:
                cmd.append(str(threads))
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, check=True)
            r = json.loads(out.stdout.strip().splitlines()[-1])
            return {'value': round(r['gib_per_s'], 4), 'unit': 'GiB/s', 'cores': r['threads'], 'kind': 'reference',
                    'host_cpus': r['cores'],
                    'sample': f'{reps} x (quantize + quantize_backward) of the full {rows}x{cols} {dname} tensor, median; '
                              f'reference fewbit/cpu path built with g++ against libtorch, {r["threads"]} intra-op thread(s) '
                              f'(the pack/unpack loops are single-threaded in the reference)',
                    'ms_per_step': round(r['seconds_per_step'] * 1e3, 2)}
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f'[bench] reference CPU baseline failed ({e}); falling back to the C port\n')
    import oracle  # the checker, timed as the CPU baseline only
    dt = DTYPES[dname]
    with np.load(tables) as z:
        borders = torch.tensor(z[f"{cfg['fn']}{bits:02d}-borders"]).to(dt)[1:-1].contiguous()
        levels = torch.tensor(z[f"{cfg['fn']}{bits:02d}-levels"]).to(dt)
    torch.manual_seed(0)
    x = torch.randn(rows, cols).to(dt)
    torch.manual_seed(1)
    gy = torch.randn(rows, cols).to(dt)
    times = []
    for i in range(4):
        t0 = time.perf_counter()
        _, state, _ = oracle.quantize(cfg['fn'], x, borders)
        oracle.quantize_backward(gy, state, levels)
        t1 = time.perf_counter()
        if i:
            times.append(t1 - t0)
    best = float(np.median(times))
    return {'value': round(nbytes / best / 2**30, 4), 'unit': 'GiB/s', 'cores': 1, 'kind': 'port',
            'sample': f'3 x (quantize + quantize_backward) of the full {rows}x{cols} {dname} tensor, median; '
                      'oracle/fewbit_oracle.c, scalar, 1 thread', 'ms_per_step': round(best * 1e3, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--config', choices=('c2', 'c4'), default='c2', help='headline workload per GPU (see module docstring)')
    ap.add_argument('--settle-ms', type=float, default=40.0,
                    help='keep issuing untimed warm-up steps until the GPU has been busy this long (0 = exactly W steps): '
                         'MI355X drops its clocks 1.5-10 ms after load begins and recovers by ~15 ms (scratch/timeline.py)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='timed region only (what the rocprofv3 passes run)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py: --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)')
        args.gpus = world
    ndev = max(torch.cuda.device_count(), 1)
    device = torch.device('cuda', local_rank % ndev)
    torch.cuda.set_device(device)
    # control plane only (barrier + max of the elapsed time): RCCL ('nccl' on ROCm).  FEWBIT_BENCH_BACKEND=gloo is a
    # validation hook that lets several ranks share one GPU, where RCCL refuses duplicate devices.
    backend = os.environ.get('FEWBIT_BENCH_BACKEND', 'nccl')
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from fewbit_amd import cabi   # raises if libfewbit_hip.so is missing: there is no fallback
    from fewbit_amd.sharding import shard_range
    cabi.lib()

    cfg = CONFIGS[args.config]
    n = cfg['rows'] * cfg['cols']
    # which elements of the (weak-scaled) global tensor this rank owns: documentation of the cut, the data is synthetic
    begin, end = shard_range(n * world, world, rank)
    assert end - begin == n
    w = Workload(cfg, device, nsets=1, seed=rank)
    fwd, bwd = w.fwd[0], w.bwd[0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_warm = time.perf_counter()
    for _ in range(args.warmup):
        fwd()
        bwd()
    # Untimed settle: a GPU that was idle boosts for ~1.5 ms, then runs 8-12 % slower for ~10 ms, then settles (measured
    # step by step with scratch/timeline.py); a 20-step region would sit in the boost phase, a 200-step region in the
    # dip.  Warm-up therefore continues (same steps, untimed) until the device has been busy for --settle-ms.
    torch.cuda.synchronize()
    settle_steps = 0
    while (time.perf_counter() - t_warm) * 1e3 < args.settle_ms:
        for _ in range(50):
            fwd()
            bwd()
        settle_steps += 50
        torch.cuda.synchronize()
    # ---- the timed region.  barrier + synchronize; an untimed pre-roll keeps the GPU busy while the host runs ahead,
    # so that the K timed steps execute from a filled queue (without it the first launch's host latency, ~5 us, is
    # 1 % of a 20-step region); then exactly K steps between two events on the launch stream.  No collective and no
    # host synchronisation inside.
    preroll = min(8, max(args.warmup, 1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(preroll):
        fwd()
        bwd()
    e0.record()
    for _ in range(args.steps):
        fwd()
        bwd()
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    barrier()
    elapsed = e0.elapsed_time(e1) * 1e-3                # seconds of GPU time for the K steps on this rank

    if world > 1:
        t = torch.tensor([elapsed, wall], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, wall = float(t[0].item()), float(t[1].item())

    if rank == 0:
        sb, fb = step_bytes(cfg)
        total = sb * args.steps * world
        value = total / elapsed / 2**30
        step_us = elapsed / args.steps * 1e6
        # per-kernel durations: K back-to-back launches of one kernel between two events (saturated queue)
        reps = max(200, min(args.steps, 2000))
        fwd_us, bwd_us = event_time_us([fwd], reps), event_time_us([bwd], reps)
        # forward's duration inside the timed region: the measured step time split in the ratio of the two kernels'
        # stand-alone durations; rocprofv3's per-dispatch average of the same command (profiles/) is the cross-check
        fwd_in_step_us = step_us * fwd_us / (fwd_us + bwd_us)
        achieved = fb / (fwd_in_step_us * 1e-6) / 1e9
        traffic, traffic_source = None, None
        tf = ROOT / 'profiles' / 'traffic_forward.json'
        if args.config == 'c2' and tf.exists():
            try:
                doc = json.loads(tf.read_text())
                traffic = doc.get('hbm_bytes_per_launch')
                traffic_source = (f"replayed from {doc.get('source', 'profiles/traffic_forward.json')}: rocprofv3 --pmc FETCH_SIZE / "
                                  "WRITE_SIZE passes over `bench.py --no-extras` (tools/profile_round.sh), NOT measured in this run")
            except Exception:  # noqa: BLE001
                traffic = None
        kernel = 'quantize_forward_lut_kernel<gelu, bf16, 3 bits>'
        line = {
            'metric': 'fwd+bwd GiB/s (and % HBM roofline) for 3-bit GELU, 4096x4096 bf16' if args.config == 'c2' else
                      'fwd+bwd GiB/s (and % HBM roofline) for 3-bit GELU, 4x(16384x4096) bf16 sharded over 8 GPUs (per-GPU shard 8192x4096)',
            'value': round(value, 2), 'unit': 'GiB/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 5), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f"{cfg['label']} per GPU, fused quantize+pack fwd / unpack+mul bwd via the C-ABI, inputs resident in HBM",
                       'name': args.config, 'elements_per_gpu': n, 'bytes_per_step_per_gpu': int(sb),
                       'parallelism': f'{world} independent shard(s) of {n} elements, cut by sharding.shard_range, no collectives',
                       'working_set_MiB_per_gpu': round(w.set_bytes / 2**20, 1),
                       'cache_state': 'warm: one buffer set re-used every step; it fits the 256 MiB Infinity Cache (see `cold`)'
                                      if w.set_bytes < INFINITY_CACHE_BYTES else 'one buffer set, larger than the 256 MiB Infinity Cache'},
            'timing': {'method': 'two HIP events on the launch stream around exactly K steps, after barrier+synchronize and an '
                                 f'untimed {preroll}-step pre-roll; max over ranks',
                       'warmup_settle': f'{settle_steps} extra untimed steps after the {args.warmup} warm-up steps, until the GPU had '
                                        f'been busy {args.settle_ms:g} ms (clock transient after idle)',
                       'wall_ms_per_step': round(wall / (args.steps + preroll) * 1e3, 5),
                       'wall_note': f'host perf_counter from the barrier to the final synchronize over K+{preroll} steps'},
            'pct_of_hbm_roofline': round(100.0 * (total / elapsed / 1e9) / (HBM_PEAK_GBS * world), 2),
            'fwd_us': round(fwd_us, 2), 'bwd_us': round(bwd_us, 2), 'fwd_in_step_us': round(fwd_in_step_us, 2),
            'roofline': {'bound': 'hbm', 'kernel': kernel, 'achieved': round(achieved, 1),
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                         'traffic': traffic, 'traffic_source': traffic_source, 'algorithmic_bytes_per_launch': int(fb),
                         'avg_launch_us': round(fwd_in_step_us, 2),
                         'avg_launch_us_method': 'timed-region step time (HIP events) x fwd/(fwd+bwd) of the back-to-back '
                                                 'per-kernel HIP-event timings (fwd_us, bwd_us)',
                         'cache_state': 'warm (x and gy are served from the Infinity Cache; writes go to HBM)'},
        }
        line['evidence'] = {'rocprofv3_kernel_stats': 'profiles/r02_bench_kernel_stats.csv (this command with --no-extras)',
                            'pmc_traffic': 'profiles/r02_pmc_traffic.json', 'per_config_rocprofv3_and_pmc': 'profiles/r02_configs.json',
                            'copy_floor_at_this_size': 'profiles/r02_stream_bench_32MiB.txt', 'regenerate': 'bash tools/profile_round.sh r02'}
        if world == 1 and not args.no_extras:
            cold = measure_config(cfg, device, cold=True)
            line['cold'] = dict(cold, note='same workload, rotating through independent buffer sets so nothing is re-used '
                                           'from L2 / Infinity Cache; frac = fwd+bwd algorithmic bytes / us_step / 8 TB/s')
            line['roofline']['frac_cold'] = round(fb / (cold['us_fwd'] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            line['roofline']['avg_launch_us_cold'] = cold['us_fwd']
            others = {}
            for name, c in CONFIGS.items():
                if name == args.config:
                    continue
                others[name] = {'workload': c['label'], 'bytes_per_step': int(step_bytes(c)[0]),
                                'warm': measure_config(c, device, cold=False)}
                if step_bytes(c)[0] < 8 * INFINITY_CACHE_BYTES:      # beyond that one buffer set is cache-cold by itself
                    others[name]['cold'] = measure_config(c, device, cold=True)
                if 'reference_published' in c:
                    others[name]['reference_published'] = c['reference_published']
                    others[name]['vs_reference_published'] = round(others[name]['warm']['GiB_s'] / c['reference_published']['GiB_s'], 2)
            line['configs'] = others
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(cfg)
            line['cpu_baseline_1thread'] = cpu_baseline(cfg, threads=1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
