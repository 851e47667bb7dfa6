#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
export SIZES=33554432,50331648,67108864,134217728,268435456
rm -f gpurun_out/r02i_headvar.log
for rep in 1 2; do
python scratch/headvar.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02i_headvar.log
FEWBIT_HIP_CHUNK=1 FEWBIT_HIP_LUT_CHUNK=2 TAGX=b1l2 python scratch/headvar.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02i_headvar.log
FEWBIT_HIP_CHUNK=1 FEWBIT_HIP_LUT_CHUNK=4 TAGX=b1l4 python scratch/headvar.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02i_headvar.log
FEWBIT_HIP_CHUNK=2 FEWBIT_HIP_LUT_CHUNK=3 TAGX=b2l3 python scratch/headvar.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02i_headvar.log
done
sort -k4,4 -s gpurun_out/r02i_headvar.log
