#!/bin/bash
# round 5, call j: why are the fp32 randomized rows slower on some boxes?  power cap / clocks of the box, the A/B with the sketch before / after the layer's GEMM
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
{ rocm-smi --showmaxpower --showpower --showclocks --showperflevel --showmemuse 2>&1 | grep -v "^=\|^$" | head -40; } > gpurun_out/r05j_smi_before.txt
timeout 900 python scratch/roberta_ab.py fp32 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05j_roberta_ab_fp32.txt
( while true; do rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk" | head -4; sleep 1; done ) > gpurun_out/r05j_smi_during.txt 2>&1 &
SMI=$!
timeout 900 python scratch/roberta_ab.py bf16 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05j_roberta_ab_bf16.txt
kill $SMI
