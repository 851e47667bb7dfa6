#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r02h_headvar.log
python scratch/headvar.py 4 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02h_headvar.log
for c in 1 2 4; do
  FEWBIT_HIP_CHUNK=$c FEWBIT_HIP_LUT_CHUNK=0 TAGX=bwdT$c python scratch/headvar.py 4 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02h_headvar.log
done
for c in 2 4 8; do
  FEWBIT_HIP_CHUNK=0 FEWBIT_HIP_LUT_CHUNK=$c TAGX=lutT$c python scratch/headvar.py 4 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02h_headvar.log
done
cat gpurun_out/r02h_headvar.log
FEWBIT_HIP_CHUNK=1 FEWBIT_HIP_LUT_CHUNK=2 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -2
