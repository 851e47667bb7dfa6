#!/bin/bash
# round 5, call l: package power / shader clock during the arms of the randomized RoBERTa step
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python scratch/roberta_power.py fp32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05l_roberta_power_fp32.txt
timeout 900 python scratch/roberta_power.py bf16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05l_roberta_power_bf16.txt
ls -la /sys/class/drm/card*/device/ 2>&1 | head -60 > gpurun_out/r05l_sysfs.txt
