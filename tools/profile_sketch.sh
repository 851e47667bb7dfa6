#!/bin/bash
# rocprofv3 passes over the random-projection kernel, settled: tools/sketch_run.py keeps the GPU busy with the same launches for
# >= 40 ms (as bench.py settles its own figures), then makes >= 200 timed launches; the summary averages the LAST 200 dispatches
# of every kernel (the settling dispatches are in the trace too and are dropped).  Kernel durations first (--kernel-trace
# --stats only), then PMC counters in their own passes (never combined with a trace domain other than --kernel-trace).  The
# program stands directly behind `--` (no env / bash -c hop under the profiler).
#   usage (through gpurun):  bash tools/profile_sketch.sh <tag> [dist rows features proj dtype [reps [settle_ms]]]
#   without a configuration: both distributions x both RoBERTa-base widths, p = 3276 of 16384 rows, bf16
#   writes  profiles/<tag>_sketch_rocprof_<dist>_<rows>x<features>_p<proj>_<dtype>.txt
set -u
TAG=${1:-r06}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; export TMPDIR=/tmp
PMC=${PMC:-1}
one() {
    local dist=$1 rows=$2 features=$3 proj=$4 dtype=$5 reps=${6:-200} settle=${7:-40}
    local name=${dist}_${rows}x${features}_p${proj}_${dtype}
    local RAW=$ROOT/gpurun_out/prof_sketch_${TAG}_$name; rm -rf "$RAW"; mkdir -p "$RAW"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/trace" -o sk -- python3 tools/sketch_run.py $dist $rows $features $proj $dtype $reps $settle > "$RAW/run.log" 2>&1
    if [ "$PMC" != 0 ]; then
        timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$RAW/pmc_sq" -o sk -- python3 tools/sketch_run.py $dist $rows $features $proj $dtype 20 0 > /dev/null 2>&1
        timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d "$RAW/pmc_tcc" -o sk -- python3 tools/sketch_run.py $dist $rows $features $proj $dtype 20 0 > /dev/null 2>&1
    fi
    python3 tools/sketch_profile_summary.py "$RAW" $reps > "profiles/${TAG}_sketch_rocprof_$name.txt" 2>&1
    mkdir -p gpurun_out/profiles_$TAG; cp "profiles/${TAG}_sketch_rocprof_$name.txt" gpurun_out/profiles_$TAG/
    head -30 "profiles/${TAG}_sketch_rocprof_$name.txt"
}
if [ $# -ge 5 ]; then
    one "$@"
else
    for features in 3072 768; do
        for dist in gaussian rademacher; do
            one $dist 16384 $features 3276 bf16 200 40
        done
    done
fi
