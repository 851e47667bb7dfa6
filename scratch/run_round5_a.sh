#!/bin/bash
# round 5, call a: generator micro-benchmark; sketch tests on the new Gaussian definition + bf16 partial sums; A/B against the round-4 generator
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 scratch/bin/gen_bench > gpurun_out/r05a_gen_bench.txt 2>&1; tail -5 gpurun_out/r05a_gen_bench.txt
timeout 900 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_linear.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05a_tests.log
P=fewbit_amd/libfewbit_hip.so; G1=scratch/libfewbit_hip_g1.so
{
for shape in "16384 768 3276" "16384 3072 3276" "16384 768 1638" "16384 3072 1638"; do
  timeout 300 python scratch/sketch_ab.py rademacher $shape new=$P new_fp32partials=$P@partials=0
  timeout 300 python scratch/sketch_ab.py gaussian $shape new=$P new_fp32partials=$P@partials=0 new_h1=$P@halves=1 new_h2=$P@halves=2 r04gen=$G1@partials=0 r04gen_h1=$G1@partials=0,halves=1 r04gen_h2=$G1@partials=0,halves=2
done
} 2>&1 | tee gpurun_out/r05a_sketch_ab.txt
