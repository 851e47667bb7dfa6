#!/bin/bash
# round 5, call ae: the width rule for fp32 input in the S-from-memory policy: GPU tests of the sketch, the layer and the bench line,
# then the arms inside fp32 RoBERTa-base
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_linear.py tests/test_gpu_bench.py -x -q -m gpu > gpurun_out/r05ae_tests.log 2>&1; tail -3 gpurun_out/r05ae_tests.log
T=$(date +%H%M%S)
{ bash scratch/box_fingerprint.sh | grep -i "vbios\|smc\|mec" | head -6; timeout 600 python scratch/roberta_ab_width.py 3 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r05ae_width_$T.txt 2>&1
cat gpurun_out/r05ae_width_$T.txt
