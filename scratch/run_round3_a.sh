#!/bin/bash
# round 3, call a: refactored launch layer (runtime tuning, describe) -- GPU suite subset, then the in-process shape sweeps
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ops.py -x -q 2>&1 | tail -3 | tee gpurun_out/r03a_tests.log
export FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_sweep.so
timeout 1500 python scratch/shape_sweep.py bwd,lut,search c2,c4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03a_shape_sweep_c2c4.txt | grep -E "^##|best"
timeout 1200 python scratch/shape_sweep.py bwd,lut,search c3k2,c3k4,f32,robbf,rob 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03a_shape_sweep_c3.txt | grep -E "^##|best"
timeout 300 python scratch/shape_sweep.py step1f,step1b c1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03a_shape_sweep_c1.txt | grep -E "^##|best"
timeout 300 scratch/stream_bench 32 > gpurun_out/r03a_stream_32.txt 2>&1; tail -8 gpurun_out/r03a_stream_32.txt
