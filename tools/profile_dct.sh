#!/bin/bash
# rocprofv3 over the sampled cosine transform (tools/dct_run.py), settled like tools/profile_sketch.sh: kernel durations (--kernel-trace --stats), then
# HBM traffic counters in their own passes.   usage (through gpurun):  bash tools/profile_dct.sh <tag>
#   writes profiles/<tag>_dct_rocprof_<rows>x<features>_p<proj>_<dtype>.txt (and a copy under gpurun_out/profiles_<tag>/): the rows a function of
#   a seed (what the layer calls); ..._explicit_idx.txt: the entry point that takes an int64 idx array
set -u
TAG=${1:-r06}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; export TMPDIR=/tmp
one() {
    local rows=$1 features=$2 proj=$3 dtype=$4 mode=${5:-seeded}
    local name=${rows}x${features}_p${proj}_${dtype}
    [ $mode = seeded ] || name=${name}_${mode}_idx
    local RAW=$ROOT/gpurun_out/prof_dct_${TAG}_$name; rm -rf "$RAW"; mkdir -p "$RAW"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/trace" -o dct -- python3 tools/dct_run.py $rows $features $proj $dtype 200 40 $mode > "$RAW/run.log" 2>&1
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$RAW/pmc_fetch" -o dct -- python3 tools/dct_run.py $rows $features $proj $dtype 20 0 $mode > /dev/null 2>&1
    timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$RAW/pmc_write" -o dct -- python3 tools/dct_run.py $rows $features $proj $dtype 20 0 $mode > /dev/null 2>&1
    python3 tools/dct_profile_summary.py "$RAW" 200 > "profiles/${TAG}_dct_rocprof_$name.txt" 2>&1
    mkdir -p gpurun_out/profiles_$TAG; cp "profiles/${TAG}_dct_rocprof_$name.txt" gpurun_out/profiles_$TAG/
    head -14 "profiles/${TAG}_dct_rocprof_$name.txt"
}
if [ $# -ge 4 ]; then one "$@"; else
    for dtype in bf16 f32; do for features in 768 3072; do one 16384 $features 3276 $dtype; done; done
    one 16384 768 3276 bf16 explicit; one 16384 3072 3276 bf16 explicit
    one 12288 768 2457 bf16                                                   # 3 x 2^12 rows: radix-3 first stage in pass B
fi
