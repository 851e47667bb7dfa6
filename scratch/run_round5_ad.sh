#!/bin/bash
# round 5, call ad (repeated): fingerprint of the box + which layer widths may read S from memory inside an fp32 model
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
T=$(date +%H%M%S)
{ bash scratch/box_fingerprint.sh | grep -i "unique_id\|vbios_version\|smc\|mec firm" | head -6; timeout 600 python scratch/roberta_ab_width.py 3 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r05ad_width_$T.txt 2>&1
cat gpurun_out/r05ad_width_$T.txt
