#!/usr/bin/env python3
"""bench.py -- forward+backward throughput of the 3-bit GELU hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: the fused quantize+pack forward and the fused unpack+mul
backward of fewbit.gelu(bits=3) over a 4096x4096 bf16 activation (BASELINE.json configs[1]), both through the C-ABI
(include/fewbit_hip.h) on HBM-resident synthetic tensors.  With N > 1 every rank runs the same per-GPU workload on
its own shard (weak scaling, no collectives on the data path; torch.distributed is only used for the barrier and
the max-over-ranks of the elapsed time).  Rank 0 prints ONE JSON line.

metric = algorithmic bytes / time, algorithmic bytes per element = 4*s + k/4 (fwd: read x, write y, write state;
bwd: read gy, read state, write gx; s = element size, k = bits) -- SURVEY.md 8(d).
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
ROWS, COLS, BITS = 4096, 4096, 3
DTYPE, DTYPE_NAME = torch.bfloat16, 'bf16'


def load_tables(device):
    with np.load(ROOT / 'fewbit_amd' / 'data' / 'builtin.npz') as z:
        borders = torch.tensor(z[f'gelu{BITS:02d}-borders']).to(DTYPE)[1:-1].contiguous().to(device)
        levels = torch.tensor(z[f'gelu{BITS:02d}-levels']).to(DTYPE).to(device)
    return borders, levels


def cpu_baseline():
    """Reference CPU path (oracle/_ref, kind 'reference') or, without it, the C restatement (kind 'port')."""
    tables = str(ROOT / 'fewbit_amd' / 'data' / 'builtin.npz')
    ref = ROOT / 'oracle' / '_ref' / 'libfewbit_ref.so'
    n = ROWS * COLS
    nbytes = n * (4 * 2 + BITS / 4)
    if ref.exists():
        reps = 10
        try:
            out = subprocess.run([sys.executable, str(ROOT / 'oracle' / 'ref_bench.py'), str(ROWS), str(COLS), DTYPE_NAME,
                                  str(BITS), str(reps), tables], capture_output=True, text=True, timeout=600, check=True)
            r = json.loads(out.stdout.strip().splitlines()[-1])
            return {'value': round(r['gib_per_s'], 4), 'unit': 'GiB/s', 'cores': r['threads'], 'kind': 'reference',
                    'sample': f'{reps} x (quantize + quantize_backward) of the full {ROWS}x{COLS} {DTYPE_NAME} tensor, median; '
                              f'reference fewbit/cpu path built with g++ against libtorch, {r["threads"]} intra-op threads '
                              f'(pack/unpack loops are single-threaded in the reference)',
                    'ms_per_step': round(r['seconds_per_step'] * 1e3, 2)}
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f'[bench] reference CPU baseline failed ({e}); falling back to the C port\n')
    import oracle
    borders, levels = load_tables('cpu')
    torch.manual_seed(0)
    x = torch.randn(ROWS, COLS).to(DTYPE)
    torch.manual_seed(1)
    gy = torch.randn(ROWS, COLS).to(DTYPE)
    times = []
    for i in range(4):
        t0 = time.perf_counter()
        _, state, _ = oracle.quantize('gelu', x, borders)
        oracle.quantize_backward(gy, state, levels)
        t1 = time.perf_counter()
        if i:
            times.append(t1 - t0)
    best = float(np.median(times))
    return {'value': round(nbytes / best / 2**30, 4), 'unit': 'GiB/s', 'cores': 1, 'kind': 'port',
            'sample': f'3 x (quantize + quantize_backward) of the full {ROWS}x{COLS} {DTYPE_NAME} tensor, median; '
                      'oracle/fewbit_oracle.c, scalar, 1 thread', 'ms_per_step': round(best * 1e3, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--no-cpu-baseline', action='store_true')
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
    cabi.lib()

    n = ROWS * COLS
    borders, levels = load_tables(device)
    # synthetic shard of this rank: seeded on the host, then resident in HBM
    torch.manual_seed(2 * rank)
    x = torch.randn(ROWS, COLS).to(DTYPE).to(device)
    torch.manual_seed(2 * rank + 1)
    gy = torch.randn(ROWS, COLS).to(DTYPE).to(device)
    y = torch.empty_like(x)
    gx = torch.empty_like(x)
    state = torch.empty(cabi.state_nbytes(n, BITS), dtype=torch.uint8, device=device)

    # pre-resolved launches: the loop below only pays ctypes + hipLaunchKernel per kernel
    fwd = cabi.bind_forward('gelu', x, borders, out=y, state=state)
    bwd = cabi.bind_backward(gy, state, levels, out=gx)

    def step():
        fwd()
        bwd()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # ---- the timed region: exactly K steps, nothing but the two launches per step on the stream
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fwd()
        bwd()
    barrier()
    elapsed = time.perf_counter() - t0

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel launch durations, HIP events on the launch stream.  Two views:
    #  (a) K back-to-back launches of one kernel between two events -> average duration per launch in a saturated
    #      queue (what roofline.achieved uses; rocprofv3's per-dispatch average in profiles/ is the cross-check);
    #  (b) the K fwd/bwd steps again with an event between every launch -> includes the few us a barrier packet
    #      plus an un-overlapped dispatch cost, reported as *_us_event_bracketed for reference only.
    def back_to_back(launch):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.steps):
            launch()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.steps

    fwd_us, bwd_us = back_to_back(fwd), back_to_back(bwd)
    nb = min(args.steps, 500)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(nb)]
    torch.cuda.synchronize()
    for i in range(nb):
        ev[i][0].record()
        fwd()
        ev[i][1].record()
        bwd()
        ev[i][2].record()
    torch.cuda.synchronize()
    fwd_us_ev = float(np.median([e[0].elapsed_time(e[1]) for e in ev])) * 1e3
    bwd_us_ev = float(np.median([e[1].elapsed_time(e[2]) for e in ev])) * 1e3

    if rank == 0:
        es = x.element_size()
        step_bytes = n * (4 * es + BITS / 4)            # 146 800 640 B
        fwd_bytes = n * (2 * es + BITS / 8)             # 73 400 320 B per forward launch
        total = step_bytes * args.steps * world
        value = total / elapsed / 2**30
        # forward's duration inside the timed region: the measured step time split in the ratio of the two kernels'
        # stand-alone (back-to-back) durations; agrees with rocprofv3's per-dispatch average of the same command
        step_us = elapsed / args.steps * 1e6
        fwd_in_step_us = step_us * fwd_us / (fwd_us + bwd_us)
        achieved = fwd_bytes / (fwd_in_step_us * 1e-6) / 1e9
        traffic = None
        tf = ROOT / 'profiles' / 'traffic_forward.json'
        if tf.exists():
            try:
                traffic = json.loads(tf.read_text()).get('hbm_bytes_per_launch')
            except Exception:  # noqa: BLE001
                traffic = None
        line = {
            'metric': 'fwd+bwd GiB/s (and % HBM roofline) for 3-bit GELU, 4096x4096 bf16',
            'value': round(value, 2), 'unit': 'GiB/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 5), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'fewbit.gelu bits={BITS} on {ROWS}x{COLS} {DTYPE_NAME} per GPU, fused quantize+pack fwd / '
                                   f'unpack+mul bwd via the C-ABI, inputs resident in HBM',
                       'elements_per_gpu': n, 'bytes_per_step_per_gpu': int(step_bytes),
                       'parallelism': f'{world} independent shard(s), no collectives'},
            'pct_of_hbm_roofline': round(100.0 * (total / elapsed / 1e9) / (HBM_PEAK_GBS * world), 2),
            'fwd_us': round(fwd_us, 2), 'bwd_us': round(bwd_us, 2), 'fwd_in_step_us': round(fwd_in_step_us, 2),
            'fwd_us_event_bracketed': round(fwd_us_ev, 2), 'bwd_us_event_bracketed': round(bwd_us_ev, 2),
            'roofline': {'bound': 'hbm', 'kernel': 'quantize_forward_lut_kernel<gelu, bf16, 3 bits>', 'achieved': round(achieved, 1),
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                         'traffic': traffic, 'algorithmic_bytes_per_launch': int(fwd_bytes),
                         'avg_launch_us': round(fwd_in_step_us, 2),
                         'avg_launch_us_method': 'timed-region step time x fwd/(fwd+bwd) of the back-to-back per-kernel '
                                                 'HIP-event timings (fwd_us, bwd_us)'},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
