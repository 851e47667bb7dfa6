#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -q -x -k "every_16bit or digests or golden" 2>&1 | tail -2
for rep in 1 2; do
for v in default lb896 lb768 lb640 lb512; do
  if [ $v = default ]; then unset FEWBIT_HIP_LIB; else export FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_$v.so; fi
  SIZES=16777216,33554432,67108864 python scratch/headvar.py 2>&1 | grep -v amdgpu.ids
done; done | sort -k4,4 -s | tee gpurun_out/r02v_lutblock.log
