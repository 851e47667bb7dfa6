#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python scratch/wall_overhead.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03e_wall_overhead.txt
S=scratch/libfewbit_hip_sweep.so
for n in 16777216 20971520 25165824 33554432 50331648; do
  SIZE=$n ROUNDS=5 timeout 300 python scratch/ablate.py bwd u2_policy=$S@u_bwd=2 u2_resident=$S@u_bwd=2,chunk=0 u2_oneshot=$S@u_bwd=2,chunk=1 u1_oneshot=$S@u_bwd=1,chunk=1 u1_resident=$S@u_bwd=1,chunk=0 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r03e_bwd_size_xover.txt
for n in 33554432 67108864; do
  SIZE=$n DT=f16 FN=silu K=2 ROUNDS=3 timeout 300 python scratch/ablate.py bwd u2_policy=$S@u_bwd=2 u1_oneshot=$S@u_bwd=1,chunk=1 2>&1 | grep -v amdgpu.ids
done | tee -a gpurun_out/r03e_bwd_size_xover.txt
for n in 16777216 50331648; do
  SIZE=$n DT=f32 ROUNDS=3 timeout 300 python scratch/ablate.py bwd u1_policy=$S@u_bwd=1 u1_oneshot=$S@u_bwd=1,chunk=1 u2_oneshot=$S@u_bwd=2,chunk=1 2>&1 | grep -v amdgpu.ids
  SIZE=$n DT=f32 ROUNDS=3 timeout 300 python scratch/ablate.py fwd u1_policy=$S@u_fwd=1 u1_oneshot=$S@u_fwd=1,chunk=1 u2_oneshot=$S@u_fwd=2,chunk=1 u1_chunk3=$S@u_fwd=1,chunk=3 2>&1 | grep -v amdgpu.ids
done | tee -a gpurun_out/r03e_bwd_size_xover.txt
