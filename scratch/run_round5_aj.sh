#!/bin/bash
# round 5, call aj: the whole GPU suite, smoke and the bench line on the final tree
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r05aj_tests.log 2>&1; tail -4 gpurun_out/r05aj_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05aj_smoke.log 2>&1; tail -2 gpurun_out/r05aj_smoke.log
python3 bench.py > gpurun_out/r05aj_bench_line.json 2> gpurun_out/r05aj_bench.err; cut -c1-300 gpurun_out/r05aj_bench_line.json
