#!/bin/bash
# round 5, call b: wave-specialisation probe; the Gaussian sketch with S written to memory once (fragments) against the fused kernel
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 scratch/bin/gen_bench > gpurun_out/r05b_gen_bench.txt 2>&1; tail -8 gpurun_out/r05b_gen_bench.txt
timeout 1200 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_linear.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05b_tests.log
P=fewbit_amd/libfewbit_hip.so
{
for shape in "16384 768 3276" "16384 3072 3276" "16384 768 1638" "16384 3072 1638" "16384 3072 8192" "65536 4096 4096"; do
  timeout 300 python scratch/sketch_ab.py gaussian $shape memory=$P@mem=1 fused=$P@mem=0 memory_w4=$P@mem=1,waves=4 rademacher_is_gaussian_arg_ignored=$P@mem=0,halves=1
  timeout 300 python scratch/sketch_ab.py rademacher $shape rademacher=$P
done
DT=f32 timeout 300 python scratch/sketch_ab.py gaussian 16384 3072 3276 memory=$P@mem=1 fused=$P@mem=0
DT=f32 timeout 300 python scratch/sketch_ab.py gaussian 16384 768 3276 memory=$P@mem=1 fused=$P@mem=0
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05b_sketch_ab.txt
