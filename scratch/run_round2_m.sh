#!/bin/bash
# full GPU suite + everything under profiles/<round>_* (tools/profile_round.sh)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/r02m_pytest.log 2>&1
tail -4 gpurun_out/r02m_pytest.log
bash tools/profile_round.sh r02 2>&1 | tail -30
