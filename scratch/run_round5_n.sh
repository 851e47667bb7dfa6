#!/bin/bash
# round 5, call n: bf16 partial sums also for the fp32 result of an fp32 input that was rounded to bf16 first: tests, stand-alone and RoBERTa A/B
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_linear.py -x -q 2>&1 | tail -8 | tee gpurun_out/r05n_tests.log
P=fewbit_amd/libfewbit_hip.so
{
for shape in "16384 768 3276" "16384 3072 3276"; do
  for dist in rademacher gaussian; do
    DT=f32 timeout 300 python scratch/sketch_ab.py $dist $shape bf16_partial_sums=$P fp32_partial_sums=$P@partials=2
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05n_sketch_ab.txt
timeout 900 python scratch/roberta_ab.py fp32 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05n_roberta_ab_fp32.txt
