#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/r02f_pytest.log 2>&1
tail -6 gpurun_out/r02f_pytest.log
python scratch/hostcost.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02f_hostcost.log
bash tools/profile_round.sh r02 2>&1 | tail -40
