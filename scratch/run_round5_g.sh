#!/bin/bash
# round 5, call g: one-launch preparation (fp32 -> bf16 copy + S fragments) against two launches: tests, stand-alone A/B, RoBERTa fp32 A/B
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sketch.py -x -q 2>&1 | tail -8 | tee gpurun_out/r05g_tests.log
P=fewbit_amd/libfewbit_hip.so
{
for shape in "16384 768 3276" "16384 3072 3276"; do
  DT=f32 timeout 300 python scratch/sketch_ab.py gaussian $shape one_launch=$P two_launches=$P@prep=0 fused=$P@mem=0
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05g_sketch_ab.txt
timeout 900 python scratch/roberta_ab.py fp32 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05g_roberta_ab_fp32.txt
