"""Where does the host wall clock of a short (K = 20 steps) timed region go beyond the GPU time of the steps?
Variants of the region's bracketing, each repeated 15 times from an idle, synchronised GPU (as bench.py's contract region)."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
import bench
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
w = bench.Workload(bench.CONFIGS['c2'], dev, nsets=1, seed=0, host_seeded=False)
fwd, bwd = w.fwd[0], w.bwd[0]
K = int(os.environ.get('K', 20))
for _ in range(200): fwd(); bwd()
torch.cuda.synchronize()

def region(mode, k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record(); torch.cuda.synchronize()
    for _ in range(5): fwd(); bwd()           # W = 5 warm-up steps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if 'e0' in mode: e0.record()
    for _ in range(k): fwd(); bwd()
    t_launched = time.perf_counter()
    if 'e1' in mode:
        e1.record()
        if 'poll' in mode:
            while not e1.query(): pass
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ev = e0.elapsed_time(e1) * 1e3 if ('e0' in mode and 'e1' in mode) else float('nan')
    return (t1 - t0) * 1e6, (t_launched - t0) * 1e6, ev

for mode in ('e0+e1+poll', 'e1+poll', 'e1', 'sync-only', 'e0+e1'):
    for k in (K, 0):
        r = [region(mode, k) for _ in range(15)]
        wall = statistics.median(x[0] for x in r); host = statistics.median(x[1] for x in r); ev = statistics.median(x[2] for x in r)
        print(f'{mode:12s} K={k:3d}: wall {wall:7.1f} us  (host done launching after {host:6.1f} us; events {ev:7.1f} us)  -> {wall / max(k, 1):6.2f} us/step', flush=True)
