#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_ops.py -q -x 2>&1 | tail -3
python scratch/step1_sizes.py 2>&1 | grep -v amdgpu.ids
FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_nosplit.so python scratch/step1_sizes.py 2>&1 | grep -v amdgpu.ids | grep float32
