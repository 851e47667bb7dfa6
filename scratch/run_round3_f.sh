#!/bin/bash
# round 3, call f: the whole GPU suite with the final build, then the round's evidence (tools/profile_round.sh r03)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r03f_tests.log
timeout 2400 bash tools/profile_round.sh r03 > gpurun_out/r03f_profile_round.log 2>&1; tail -30 gpurun_out/r03f_profile_round.log
ls gpurun_out/profiles_r03 | head -50
