#!/bin/bash
# Why does the S-from-memory path slow the OTHER kernels of an fp32 RoBERTa step down on most leases?  rocprofv3 --pmc (one counter
# group per pass, program directly after --) over one randomized row of tools/roberta_bench.py --table, once with the fused Gaussian
# kernel (FEWBIT_SKETCH_MATERIALISE=0) and once with S from memory (=1); per kernel class: duration and counters per step.
#   usage (through gpurun): bash tools/profile_insitu_pmc_roberta.sh <tag>
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; export TMPDIR=/tmp
RAW=$ROOT/gpurun_out/insitu_pmc_roberta_$TAG; rm -rf "$RAW"; mkdir -p "$RAW"
i=0
for group in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for arm in 0 1; do
    export FEWBIT_SKETCH_MATERIALISE=$arm
    timeout 600 rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$RAW/g${i}_arm$arm" -o p -- python3 tools/roberta_bench.py --table --row 2 --dtype fp32 --matmul gaussian --steps 3 > "$RAW/g${i}_arm$arm.log" 2>&1 || echo "group $i ($group) arm $arm failed" >> "$RAW/failed.txt"
  done
done
unset FEWBIT_SKETCH_MATERIALISE
python3 - "$RAW" <<'PY' > "$ROOT/gpurun_out/insitu_pmc_roberta_$TAG.txt"
import csv, glob, sys, collections
sys.path.insert(0, 'tools')
from kernel_classes import classify
raw = sys.argv[1]
print('# tools/profile_insitu_pmc_roberta.sh: RoBERTa-base fp32, every encoder Linear randomized (Gaussian, ratio 0.2), 3 + 3 warm-up steps + the memory-usage pass = 7 steps per process;')
print('# per kernel class: total over the process of the counter, and of the kernel durations in that same pass (ms); arm 0 = fused Gaussian kernel, arm 1 = S from memory')
rows = collections.defaultdict(dict)
for d in sorted(glob.glob(raw + '/g*_arm*')):
    if not d[-1].isdigit(): continue
    arm = d[-1]
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); seen = set()
        for r in csv.DictReader(open(f)):
            c = classify(r['Kernel_Name'])
            acc[r['Counter_Name']][c] += float(r['Counter_Value'])
            if (r['Dispatch_Id'], r['Counter_Name']) not in seen and r['Counter_Name'] == next(iter(acc)):
                dur[c] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        for counter, per in acc.items():
            for c, v in per.items():
                rows[(counter, c)][arm] = (v, dur[c])
print(f'{"counter":34s} {"kernel class":22s} {"fused: value":>16s} {"ms":>9s} {"from memory: value":>20s} {"ms":>9s} {"ratio":>7s}')
for (counter, c), d in sorted(rows.items()):
    a, b = d.get('0', (0, 0)), d.get('1', (0, 0))
    print(f'{counter:34s} {c:22s} {a[0]:16.0f} {a[1]:9.2f} {b[0]:20.0f} {b[1]:9.2f} {(b[0] / a[0] if a[0] else float("nan")):7.3f}')
try:
    print(open(raw + '/failed.txt').read())
except OSError:
    pass
PY
find "$RAW" -name "*.csv" -delete          # (the raw per-dispatch rows are ~100 MB; the summary above is what travels back)
cat "$ROOT/gpurun_out/insitu_pmc_roberta_$TAG.txt" | head -70
