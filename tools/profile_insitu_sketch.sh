#!/bin/bash
# Where a RoBERTa-base step with randomized linear layers spends its GPU time: rocprofv3 kernel stats of single rows of
# tools/roberta_bench.py --table, grouped by tools/kernel_classes.py  ->  gpurun_out/<tag>_roberta_randomized_insitu.json
set -u
R=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; export TMPDIR=/tmp
RAW=$ROOT/gpurun_out/insitu_sketch_$R; rm -rf "$RAW"; mkdir -p "$RAW"
OUT=$ROOT/gpurun_out/${R}_roberta_randomized_insitu.json
echo "[" > "$OUT"; first=1
for v in "bf16 0 rademacher" "bf16 2 rademacher" "bf16 2 gaussian" "bf16 2 dct" "fp32 0 gaussian" "fp32 2 gaussian" "fp32 2 rademacher" "fp32 2 dct"; do
    set -- $v; tag=$1_row$2_$3
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/$tag" -o rob -- \
        python3 tools/roberta_bench.py --table --row $2 --dtype $1 --matmul $3 --steps 6 > "$RAW/$tag.json" 2> "$RAW/$tag.err"
    find "$RAW/$tag" -name "*kernel_trace.csv" -delete
    [ $first = 1 ] || echo "," >> "$OUT"; first=0
    python3 tools/kernel_classes.py "$RAW/$tag" 10 "$1, GELU vanilla, linear $([ $2 = 0 ] && echo vanilla || echo "randomized ($3)")" >> "$OUT"
done
echo "]" >> "$OUT"
cat "$OUT"
