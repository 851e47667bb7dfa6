#!/usr/bin/env python3
"""Time the REFERENCE's own CPU path on a bounded sample -- the `cpu_baseline` leg of bench.py.

  table configs (gelu / silu tables)  oracle/_ref/libfewbit_ref.so = the reference's fewbit.cc + cpu/gelu.cc + cpu/codec.cc
                                      built with g++ against this image's libtorch: torch.ops.fewbit.quantize then
                                      quantize_backward (fewbit/cpu/gelu.cc:7-45).  NB the reference's quantize always
                                      evaluates torch::gelu, whatever table it is given.
  1-bit config (relu)                 the reference's native quantize segfaults for a 1-bit table (SURVEY 2.2 defect 5), so --
                                      as BASELINE.md section 2 did -- the reference's bit codec Deflate/Inflate(..., 1)
                                      (oracle/_ref/libcodec_ref.so = fewbit/cpu/codec.h:33-83 compiled as is, single
                                      thread) around ATen relu / compare / multiply.

TEST/BENCH INFRASTRUCTURE: runs in its own process because the reference library registers the same
TORCH_LIBRARY(fewbit) namespace as the product's libfewbit.so.  Prints one JSON line.
usage: ref_bench.py ROWS COLS DTYPE BITS REPS TABLES_NPZ [THREADS (0 = torch default)] [FN (gelu)]
"""
import ctypes
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
    if len(sys.argv) > 7 and int(sys.argv[7]) > 0:
        torch.set_num_threads(int(sys.argv[7]))
    fn = sys.argv[8] if len(sys.argv) > 8 else 'gelu'
    dtype = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[dtype_name]
    torch.manual_seed(0)
    x = torch.randn(rows, cols).to(dtype)
    torch.manual_seed(1)
    gy = torch.randn(rows, cols).to(dtype)
    n = rows * cols

    if fn == 'relu':
        codec = ctypes.CDLL(str(HERE / '_ref' / 'libcodec_ref.so'))
        codec.ref_deflate_u8.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int32]
        codec.ref_inflate_u8.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int32]

        def step():
            y = torch.relu(x)
            codes = (x > 0).to(torch.int32).reshape(-1)                   # the bit rule of fewbit/cuda/codec.cu:412-425
            state = torch.empty((n + 7) // 8, dtype=torch.uint8)
            codec.ref_deflate_u8(codes.data_ptr(), n, state.data_ptr(), 1)   # fewbit/cpu/codec.h:33-57
            back = torch.empty(n, dtype=torch.int32)
            codec.ref_inflate_u8(back.data_ptr(), n, state.data_ptr(), 1)    # fewbit/cpu/codec.h:59-83
            gx = back.reshape(rows, cols).to(dtype) * gy
            return y, state, gx
        what = 'relu + reference Deflate(...,1) | reference Inflate(...,1) + multiply'
    else:
        torch.ops.load_library(str(HERE / '_ref' / 'libfewbit_ref.so'))
        with np.load(tables) as z:
            borders = torch.tensor(z[f'{fn}{bits:02d}-borders']).to(dtype)[1:-1].contiguous()
            levels = torch.tensor(z[f'{fn}{bits:02d}-levels']).to(dtype)

        def step():
            y, state = torch.ops.fewbit.quantize(x, borders)              # fewbit/cpu/gelu.cc:7-31
            gx = torch.ops.fewbit.quantize_backward(gy, state, levels)    # fewbit/cpu/gelu.cc:33-45
            return y, state, gx
        what = f'reference quantize + quantize_backward, {fn} {bits}-bit table'

    times = []
    for i in range(reps + 1):
        t0 = time.perf_counter()
        y, state, gx = step()
        t1 = time.perf_counter()
        if i:                                                         # first pass is warm-up
            times.append(t1 - t0)
    nbytes = n * (4 * x.element_size() + bits / 4)
    best = float(np.median(times))
    print(json.dumps({'seconds_per_step': best, 'gib_per_s': nbytes / best / 2**30, 'threads': torch.get_num_threads(),
                      'cores': os.cpu_count(), 'reps': reps, 'what': what, 'checksum': int(state.sum().item())}))


if __name__ == '__main__':
    main()
