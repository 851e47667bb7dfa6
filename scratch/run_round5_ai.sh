#!/bin/bash
# round 5, call ai (repeated): the fragment launch in FRONT of the conversion pass of an fp32 input (fragfirst) against behind it (production), all arms
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
T=$(date +%H%M%S)
OUT=gpurun_out/r05ai_fragfirst_$T.txt
bash scratch/box_fingerprint.sh | grep -i "vbios_version\|smc\|MEC firm" | head -4 > $OUT
for v in ${VARIANTS:-production fragfirst production fragfirst}; do
    echo "== $v" >> $OUT
    if [ $v = production ]; then unset FEWBIT_HIP_LIB; else export FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_$v.so; fi
    timeout 300 python scratch/roberta_ab_width.py 2 2>&1 | grep -v amdgpu.ids >> $OUT
done
cat $OUT
