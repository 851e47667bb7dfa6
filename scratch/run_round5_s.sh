#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scratch/dbg_fuzz2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05s_dbg.txt
bash scratch/run_round5_q.sh
