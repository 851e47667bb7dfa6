#!/bin/bash
# round 5, call aa: which of the from-memory path's kernels costs the fp32 model's GEMMs their clock -- the VALU-only fragment kernel or the product kernel that reads it?
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python scratch/roberta_ab.py fp32 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05aa_roberta_ab_fp32.txt
