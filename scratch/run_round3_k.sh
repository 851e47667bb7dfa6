#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_sweep.so timeout 1500 python scratch/shape_sweep.py step1f,step1b relu16_2m,relu16_4m,relu16_8m,relu16_12m,relu16_25m,relu32_8m,relu32_16m 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03k_shape_sweep_step1_sizes.txt | grep -E "^##|best"
