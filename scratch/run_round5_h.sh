#!/bin/bash
# round 5, call h: the whole GPU suite, smoke, the default bench line
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r05h_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r05h_smoke.log
timeout 900 python bench.py > gpurun_out/r05h_bench_line.json 2> gpurun_out/r05h_bench.err; tail -c 1500 gpurun_out/r05h_bench_line.json
