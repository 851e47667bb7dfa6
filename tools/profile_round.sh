#!/bin/bash
# Collect the per-round rocprofv3 evidence for bench.py on the GPU box and summarise it into profiles/.
#   usage (through gpurun):  bash tools/profile_round.sh r01
# Three separate passes, as MI355X_MICROARCH.md prescribes: kernel trace + stats, PMC FETCH_SIZE, PMC WRITE_SIZE.
# Raw output goes to gpurun_out/prof_<round>/ (scratch); the summaries are written to gpurun_out/profiles_<round>/
# and are copied into profiles/ by hand after review.
set -u
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export TMPDIR=/tmp
RAW=$ROOT/gpurun_out/prof_$R
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p "$RAW" "$OUT"
python3 bench.py > "$OUT/${R}_bench_line.json" 2> "$RAW/bench.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/trace" -o bench -- python3 bench.py --no-cpu-baseline > "$RAW/bench_under_trace.json" 2> "$RAW/trace.err"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$RAW/fetch" -o bench -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> "$RAW/fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$RAW/write" -o bench -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> "$RAW/write.err"
python3 tools/summarize_profiles.py "$R" "$RAW" "$OUT"
ls -la "$OUT"
