#!/bin/bash
# round 3, call q: the exhaustive sweeps once more on the final build (new tile widths / launch policy)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python scratch/all_fp32_patterns.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03q_all_fp32_patterns.txt
timeout 1500 python scratch/all_patterns_more.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r03q_all_patterns_more.txt
timeout 2400 python scratch/ragged_sweep.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r03q_ragged_sweep.txt
