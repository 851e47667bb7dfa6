#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
export SIZES=25165824,33554432,50331648,67108864,268435456
rm -f gpurun_out/r02q_headvar.log
for spec in "-1 -1" "1 1" "1 2" "1 3" "2 5" "3 7" "1 4"; do set -- $spec
  FEWBIT_HIP_CHUNK=$1 FEWBIT_HIP_LUT_CHUNK=$2 TAGX=b$1l$2 python scratch/headvar.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02q_headvar.log
done
sort -k4,4 -s gpurun_out/r02q_headvar.log
