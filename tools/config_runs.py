#!/usr/bin/env python3
"""Run every BASELINE config (bench.CONFIGS) cache-warm and cache-cold in a fixed, recorded order of launches, so that a
rocprofv3 per-dispatch trace (kernel trace or PMC) of this program can be split back into phases by launch count.

    python3 tools/config_runs.py PHASES_JSON [launches-per-phase]
    rocprofv3 --kernel-trace --output-format csv -d DIR -o cfg -- python3 tools/config_runs.py DIR/phases.json

Every launch goes through the C-ABI exactly as in bench.py.  PHASES_JSON lists, in launch order, the phases
{config, mode, launches, kernel order 'fwd,bwd,fwd,bwd,...'}; tools/summarize_profiles.py pairs it with the trace."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import bench  # noqa: E402


SETTLE_S = 0.04


def main():
    out = Path(sys.argv[1])
    per_phase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    device = torch.device('cuda', 0)
    torch.cuda.set_device(device)
    phases = []
    for name, cfg in bench.CONFIGS.items():
        big = bench.step_bytes(cfg)[0] >= 8 * bench.INFINITY_CACHE_BYTES      # one buffer set is cache-cold by itself
        for mode in (('warm', ) if big else ('warm', 'cold')):
            nsets = bench.nsets_for_cold(cfg) if mode == 'cold' else 1
            w = bench.Workload(cfg, device, nsets=nsets, seed=3, host_seeded=False)
            steps = w.steps()
            torch.cuda.synchronize()
            for f in steps:                                # first touch of every buffer set
                f()
            torch.cuda.synchronize()
            launches = len(steps)
            # untimed settle: the GPU's clocks dip 1.5-10 ms after load begins (scratch/timeline.py); run the same
            # launches until the device has been busy ~40 ms, counted into the same 'touch' phase
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < SETTLE_S:
                for f in steps:
                    f()
                launches += len(steps)
                torch.cuda.synchronize()
            phases.append({'config': name, 'mode': 'touch', 'launches': launches})
            rounds = max(2, per_phase // nsets)
            for _ in range(rounds):
                for f in steps:
                    f()
            torch.cuda.synchronize()
            sb, fb = bench.step_bytes(cfg)
            phases.append({'config': name, 'mode': mode, 'launches': rounds * len(steps), 'buffer_sets': nsets,
                           'algorithmic_bytes_per_launch': int(fb), 'workload': cfg['label']})
            del w, steps
            torch.cuda.empty_cache()
    out.parent.mkdir(parents=True, exist_ok=True)
    out.write_text(json.dumps({'order': 'fwd,bwd alternating inside every phase', 'phases': phases}, indent=1))


if __name__ == '__main__':
    main()
