#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
S=scratch/libfewbit_hip
for w in fwd bwd step; do ROUNDS=9 timeout 300 python scratch/ablate.py $w r02=${S}_r02.so prod=${S}_prod.so sweep=${S}_sweep.so r02b=${S}_r02.so 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03g_r02_vs_r03.txt
for n in 33554432 50331648; do
SIZE=$n ROUNDS=5 timeout 600 python scratch/ablate.py fwd base=${S}_sweep.so noact=${S}_lutA1.so nolookup=${S}_lutA2.so noact_nolookup=${S}_lutA3.so nobuild_nolookup=${S}_lutA6.so nostate=${S}_lutA8.so copyonly=${S}_lutA15.so ntstate=${S}_lutA16.so 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r03g_ablate_fwd_large.txt
ROUNDS=5 timeout 600 python scratch/ablate.py step base=${S}_sweep.so ntstate=${S}_lutA16.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03g_ablate_fwd_large.txt
