#!/bin/bash
# round 5, call u: woven generator (next step's fragment generated between this step's MFMAs, -DFEWBIT_GAUSSIAN_WOVEN=1) against the clump: correctness, then A/B
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_woven.so timeout 1200 python -m pytest tests/test_gpu_sketch.py -x -q -k "product or fuzz or fragments or slices" 2>&1 | tail -3 | tee gpurun_out/r05u_tests.log
P=fewbit_amd/libfewbit_hip.so; WV=scratch/libfewbit_hip_woven.so
{
for shape in "16384 768 3276" "16384 3072 3276" "16384 768 1638"; do
  timeout 300 python scratch/sketch_ab.py gaussian $shape clump_h1=$P@mem=0,halves=1 woven_h1=$WV@mem=0,halves=1 clump_w4=$P@mem=0,halves=1,waves=4 woven_w4=$WV@mem=0,halves=1,waves=4 memory=$P@mem=1
done
DT=f32 timeout 300 python scratch/sketch_ab.py gaussian 16384 768 3276 clump=$P woven=$WV
DT=f32 timeout 300 python scratch/sketch_ab.py gaussian 16384 3072 3276 clump_h1=$P@halves=1 woven_h1=$WV@halves=1 clump_policy=$P
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05u_sketch_ab.txt
