#!/bin/bash
# round 6, call e: the round's evidence on the tree with the sampled-DCT kernel pair (tools/profile_round.sh r06), after the whole GPU suite
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 | cut -c1-300 > gpurun_out/r06e_pytest_gpu.txt; cat gpurun_out/r06e_pytest_gpu.txt
bash tools/profile_round.sh r06 > gpurun_out/r06e_profile_round.log 2>&1; tail -5 gpurun_out/r06e_profile_round.log
