#!/bin/bash
# round 6, call o: experiment: pass B serving its samples with 16 (shipped) / 8 / 4 lanes per sample
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in serve8 serve4; do
  FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_dct_$v.so timeout 600 python3 -m pytest tests/test_gpu_dct.py -q -m gpu -x 2>&1 | tail -2 | cut -c1-300
done
MODE=seeded VARIANTS="prod serve16 serve8 serve4 prod serve8 serve4" bash scratch/run_round6_d.sh | tail -21
