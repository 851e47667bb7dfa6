#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
export SIZES=8388608,12582912,16777216,25165824
for rep in 1 2; do
for spec in "-1 -1" "1 -1"; do set -- $spec
  FEWBIT_HIP_CHUNK=$1 FEWBIT_HIP_LUT_CHUNK=$2 TAGX=b$1 python scratch/headvar.py 2>&1 | grep -v amdgpu.ids
done; done | sort -k4,4 -s | tee gpurun_out/r02r_headvar.log
DT=f32 SIZES=4194304,8388608,16777216 FEWBIT_HIP_CHUNK=1 TAGX=f32b1 python scratch/headvar.py 2>&1 | grep -v amdgpu.ids
DT=f32 SIZES=4194304,8388608,16777216 TAGX=f32auto python scratch/headvar.py 2>&1 | grep -v amdgpu.ids
