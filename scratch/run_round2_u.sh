#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
for w in 32 28 24 20 16; do
  FEWBIT_HIP_WAVES_PER_CU=$w TAGX=w$w SIZES=16777216,33554432 python scratch/headvar.py 2>&1 | grep -v amdgpu.ids
done | sort -k4,4 -s | tee gpurun_out/r02u_waves.log
