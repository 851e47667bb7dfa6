#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_sweep.so timeout 1500 python scratch/shape_sweep.py bwd,search,lut f32_4m,f32_8m,bf16_4m,bf16_8m,bf16_12m 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03l_shape_sweep_mid_sizes.txt | grep -E "^##|best"
FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_sweep.so timeout 600 python scratch/shape_sweep.py step1f,step1b relu16c2,relu16_8m,relu32_8m 2>&1 | grep -v amdgpu.ids | grep -E "^##"
