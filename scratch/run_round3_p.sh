#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
S=scratch/libfewbit_hip
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or every_16bit or full_size_digests" 2>&1 | tail -2
for w in fwd bwd step; do ROUNDS=9 timeout 300 python scratch/ablate.py $w base=${S}_sweep.so saddr=${S}_saddr.so base2=${S}_sweep.so saddr2=${S}_saddr.so 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03p_saddr.txt
SIZE=33554432 ROUNDS=5 timeout 300 python scratch/ablate.py step base=${S}_sweep.so saddr=${S}_saddr.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03p_saddr.txt
