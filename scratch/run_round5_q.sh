#!/bin/bash
# round 5, call q (repeated): fingerprint of the box + the two-arm classifier (does S from memory help or hurt an fp32 model here?)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
T=$(date +%H%M%S)
{ bash scratch/box_fingerprint.sh; timeout 600 python scratch/roberta_ab_short.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r05q_box_$T.txt 2>&1
tail -1 gpurun_out/r05q_box_$T.txt
