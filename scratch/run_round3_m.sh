#!/bin/bash
# round 3, call m: final binaries (split build, final policies) -- GPU suite, the round's evidence once more
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r03m_tests.log
timeout 2400 bash tools/profile_round.sh r03 > gpurun_out/r03m_profile_round.log 2>&1; tail -22 gpurun_out/r03m_profile_round.log | head -14
S=scratch/libfewbit_hip
for w in fwd bwd step; do ROUNDS=9 timeout 300 python scratch/ablate.py $w r02=${S}_r02.so prod=${S}_prod.so r02b=${S}_r02.so prodb=${S}_prod.so 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03m_r02_vs_r03.txt
