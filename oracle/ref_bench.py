#!/usr/bin/env python3
"""Time the REFERENCE's own CPU path (oracle/_ref/libfewbit_ref.so = fewbit.cc + cpu/gelu.cc + cpu/codec.cc built
with g++ against this image's libtorch) on a bounded sample -- the `cpu_baseline` leg of bench.py.

TEST/BENCH INFRASTRUCTURE: runs in its own process because the reference library registers the same
TORCH_LIBRARY(fewbit) namespace as the product's libfewbit.so.  Prints one JSON line.
usage: ref_bench.py ROWS COLS DTYPE BITS REPS TABLES_NPZ [THREADS]
"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent


def main():
    rows, cols, dtype_name, bits, reps, tables = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), \
        int(sys.argv[5]), sys.argv[6]
    if len(sys.argv) > 7:
        torch.set_num_threads(int(sys.argv[7]))
    dtype = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[dtype_name]
    so = HERE / '_ref' / 'libfewbit_ref.so'
    torch.ops.load_library(str(so))
    with np.load(tables) as z:
        borders = torch.tensor(z[f'gelu{bits:02d}-borders']).to(dtype)[1:-1].contiguous()
        levels = torch.tensor(z[f'gelu{bits:02d}-levels']).to(dtype)
    torch.manual_seed(0)
    x = torch.randn(rows, cols).to(dtype)
    torch.manual_seed(1)
    gy = torch.randn(rows, cols).to(dtype)
    times = []
    for i in range(reps + 1):
        t0 = time.perf_counter()
        y, state = torch.ops.fewbit.quantize(x, borders)              # fewbit/cpu/gelu.cc:7-31
        gx = torch.ops.fewbit.quantize_backward(gy, state, levels)    # fewbit/cpu/gelu.cc:33-45
        t1 = time.perf_counter()
        if i:                                                         # first pass is warm-up
            times.append(t1 - t0)
    n = rows * cols
    nbytes = n * (4 * x.element_size() + bits / 4)
    best = float(np.median(times))
    print(json.dumps({'seconds_per_step': best, 'gib_per_s': nbytes / best / 2**30, 'threads': torch.get_num_threads(),
                      'cores': os.cpu_count(), 'reps': reps, 'checksum': int(state.sum().item())}))


if __name__ == '__main__':
    main()
