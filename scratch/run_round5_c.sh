#!/bin/bash
# round 5, call c: sketch tests; XCD-aware tile order and fragment prefetch depth A/B; rocprofv3 of the S-from-memory path
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_linear.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05c_tests.log
P=fewbit_amd/libfewbit_hip.so; NX=scratch/libfewbit_hip_noxcd.so; A8=scratch/libfewbit_hip_ahead8.so; A2=scratch/libfewbit_hip_ahead2.so
{
for shape in "16384 768 3276" "16384 3072 3276" "16384 3072 1638" "65536 4096 4096"; do
  timeout 300 python scratch/sketch_ab.py gaussian $shape memory=$P@mem=1 memory_noxcd=$NX@mem=1 memory_ahead8=$A8@mem=1 memory_ahead2=$A2@mem=1 fused=$P@mem=0 fused_noxcd=$NX@mem=0
  timeout 300 python scratch/sketch_ab.py rademacher $shape rademacher=$P rademacher_noxcd=$NX
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05c_sketch_ab.txt
bash tools/profile_sketch.sh r05c_gaussian_3072 gaussian 16384 3072 3276 bf16 10 > gpurun_out/r05c_prof_gaussian_3072.txt 2>&1
bash tools/profile_sketch.sh r05c_gaussian_768 gaussian 16384 768 3276 bf16 10 > gpurun_out/r05c_prof_gaussian_768.txt 2>&1
bash tools/profile_sketch.sh r05c_rademacher_3072 rademacher 16384 3072 3276 bf16 10 > gpurun_out/r05c_prof_rademacher_3072.txt 2>&1
head -12 gpurun_out/r05c_prof_gaussian_3072.txt
