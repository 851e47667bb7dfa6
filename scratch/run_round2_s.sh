#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_numerics.py tests/test_gpu_ops.py -q -x 2>&1 | tail -3
for v in default nosplit; do
  if [ $v = default ]; then unset FEWBIT_HIP_LIB; else export FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_$v.so; fi
  DT=f32 SIZES=1048576,4194304,16777216,50331648,134217728 python scratch/headvar.py 2>&1 | grep -v amdgpu.ids
done | sort -k4,4 -s | tee gpurun_out/r02s_headvar.log
