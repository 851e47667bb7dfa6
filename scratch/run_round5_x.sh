#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sketch.py -x -q -k "half_a_gigabyte" 2>&1 | tail -15 | cut -c1-400 | tee gpurun_out/r05x_tests.log
