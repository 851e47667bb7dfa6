#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
scratch/stream_bench 32 2>&1 | tee gpurun_out/r02d_stream32.log
scratch/stream_bench 128 2>&1 | tee gpurun_out/r02d_stream128.log
for a in "20 5" "200 5" "200 20" "2000 50"; do set -- $a; python bench.py --steps $1 --warmup $2 --no-extras --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench K=$1 W=$2', d['value'], d['ms_per_step'], d['timing']['wall_ms_per_step'], d['fwd_us'], d['bwd_us'], d['timing']['warmup_settle'][:20])"; done
