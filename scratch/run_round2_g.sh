#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/profile_round.sh r02 2>&1 | tail -16
