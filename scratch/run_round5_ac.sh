#!/bin/bash
# round 5, call ac: Rademacher with the next block's Philox call spread one round per MFMA slot over every stage (-DFEWBIT_RADEMACHER_WOVEN=1) against the call as one clump per block
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_radwoven.so timeout 1200 python -m pytest tests/test_gpu_sketch.py -x -q -k "product or fuzz or slices or rademacher" 2>&1 | tail -3 | tee gpurun_out/r05ac_tests.log
P=fewbit_amd/libfewbit_hip.so; RW=scratch/libfewbit_hip_radwoven.so
{
for shape in "16384 768 3276" "16384 3072 3276" "16384 768 1638" "16384 3072 1638" "65536 4096 4096"; do
  timeout 300 python scratch/sketch_ab.py rademacher $shape clump=$P woven=$RW clump_w4=$P@waves=4 woven_w4=$RW@waves=4
done
DT=f32 timeout 300 python scratch/sketch_ab.py rademacher 16384 768 3276 clump=$P woven=$RW
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05ac_sketch_ab.txt
