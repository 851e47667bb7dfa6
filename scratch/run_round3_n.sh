#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
S=scratch/libfewbit_hip
for w in fwd bwd step; do ROUNDS=9 timeout 300 python scratch/ablate.py $w base=${S}_sweep.so prio1=${S}_prio1.so prio3=${S}_prio3.so early=${S}_early.so base2=${S}_sweep.so 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03n_setprio.txt
SIZE=33554432 ROUNDS=5 timeout 300 python scratch/ablate.py step base=${S}_sweep.so prio1=${S}_prio1.so prio3=${S}_prio3.so early=${S}_early.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03n_setprio.txt
SIZE=16777216 DT=f32 ROUNDS=5 timeout 300 python scratch/ablate.py step base=${S}_sweep.so prio1=${S}_prio1.so prio3=${S}_prio3.so early=${S}_early.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03n_setprio.txt
