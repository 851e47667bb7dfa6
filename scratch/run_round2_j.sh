#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
export SIZES=16777216,33554432,50331648,67108864,134217728,268435456
rm -f gpurun_out/r02j_headvar.log
python scratch/headvar.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02j_headvar.log
for b in 1 3 5; do for l in 3 5 6 7 12; do
  case "$b$l" in 13|35|56|17|312|15|37) FEWBIT_HIP_CHUNK=$b FEWBIT_HIP_LUT_CHUNK=$l TAGX=b${b}l$l python scratch/headvar.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02j_headvar.log;; esac
done; done
sort -k4,4 -s gpurun_out/r02j_headvar.log
export DT=f32 SIZES=16777216,50331648,134217728
for b in -1 1 2 3 5; do FEWBIT_HIP_CHUNK=$b TAGX=f32b$b python scratch/headvar.py 2>&1 | grep -v amdgpu.ids; done | sort -k4,4 -s | tee gpurun_out/r02j_headvar_f32.log
