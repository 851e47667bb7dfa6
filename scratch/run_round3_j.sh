#!/bin/bash
# round 3, call j: soak fuzz with randomised launch shapes, new large-fp32 parity test, 1-bit family shape sweep at large sizes
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "large_fp32 or every_launch_shape or describe" 2>&1 | tail -3
FEWBIT_FUZZ_SEEDS=300 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3 | tee gpurun_out/r03j_soak_fuzz.txt
FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_sweep.so timeout 900 python scratch/shape_sweep.py step1f,step1b relu16c2,relu16,relu32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03j_shape_sweep_step1.txt | grep -E "^##|best"
