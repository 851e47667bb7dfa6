#!/bin/bash
# round 5, call ar: the width arms on a RoBERTa-large-shaped fp32 model (1024 / 4096 wide, 24 layers, batch 128 x seq 128): where does the width rule's threshold belong?
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
T=$(date +%H%M%S)
{ bash scratch/box_fingerprint.sh | grep -i "vbios_version\|smc\|MEC firm" | head -4; LARGE=1 timeout 1200 python scratch/roberta_ab_width.py 2 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r05ar_width_large_$T.txt 2>&1
cat gpurun_out/r05ar_width_large_$T.txt
