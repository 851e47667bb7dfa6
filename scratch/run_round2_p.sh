#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -2
SIZES=16777216,33554432,67108864,268435456 python scratch/headvar.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02p_headvar.log
DT=f32 SIZES=16777216,50331648 python scratch/headvar.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02p_headvar.log
