#!/bin/bash
# round 5, call f: RoBERTa-base step, arms interleaved in one process (is the S-from-memory path slower IN the model than the fused kernel?)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python scratch/roberta_ab.py fp32 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05f_roberta_ab_fp32.txt
timeout 900 python scratch/roberta_ab.py bf16 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05f_roberta_ab_bf16.txt
