#!/bin/bash
# rocprofv3 over scratch/insitu_pmc.py: kernel durations, then TCC counters one group per pass (never with other trace domains
# than --kernel-trace).  Summary -> gpurun_out/insitu_pmc_<tag>.txt      usage (through gpurun): bash tools/profile_insitu_pmc.sh <tag>
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; export TMPDIR=/tmp
RAW=$ROOT/gpurun_out/insitu_pmc_$TAG; rm -rf "$RAW"; mkdir -p "$RAW"
rocprofv3 -L > "$RAW/counters_available.txt" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$RAW/trace" -o p -- python3 scratch/insitu_pmc.py > "$RAW/trace.log" 2>&1
i=0
for group in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum" "WRITE_SIZE" "FETCH_SIZE" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum" "TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
             "TCC_WRITE_sum TCC_READ_sum" "TCC_EA0_ATOMIC_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum" "GRBM_GUI_ACTIVE" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" \
             "TCC_NORMAL_EVICT_sum TCC_ALL_TC_OP_INV_EVICT_sum" "TCC_BUSY_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$RAW/pmc_$i" -o p -- python3 scratch/insitu_pmc.py > "$RAW/pmc_$i.log" 2>&1 || echo "pass $i ($group) failed" >> "$RAW/failed.txt"
done
python3 - "$RAW" <<'PY' > "$ROOT/gpurun_out/insitu_pmc_$TAG.txt"
import csv, glob, json, sys, collections, statistics
raw = sys.argv[1]
manifest = None
for line in open(raw + '/trace.log'):
    if line.startswith('MANIFEST '):
        manifest = json.loads(line[9:])
def classify(name):
    if 'quantize_forward' in name: return 'fwd'
    if 'neg_kernel' in name: return 'neg'
    return None
def split(rows_by_class):
    """dispatch rows (in time order) of the two kernel classes -> {(kernel, state): rows}"""
    out, pos = {}, {'fwd': 0, 'neg': 0}
    for m in manifest:
        cls = 'neg' if m['kernel'] == 'neg_in_place' else 'fwd'
        out[(m['kernel'], m['state'])] = rows_by_class[cls][pos[cls]:pos[cls] + m['dispatches']]
        pos[cls] += m['dispatches']
    return out
print('# scratch/insitu_pmc.py under rocprofv3: per (kernel, state) the MEDIAN over the last 20 of 30 dispatches; 16384 x 3072 bf16')
# durations
for f in glob.glob(raw + '/trace/**/*kernel_trace.csv', recursive=True):
    rows = collections.defaultdict(list)
    for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp'])):
        c = classify(r['Kernel_Name'])
        if c: rows[c].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in split(rows).items():
        print(f'duration_us            {k[0]:18s} {k[1]:22s} {statistics.median(v[10:]):10.2f}')
for d in sorted(glob.glob(raw + '/pmc_*')):
    if not d.split('_')[-1].isdigit(): continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in sorted(csv.DictReader(open(f)), key=lambda r: (int(r['Dispatch_Id']))):
            c = classify(r['Kernel_Name'])
            if c: per[r['Counter_Name']][c].append(float(r['Counter_Value']))
        for counter, rows in per.items():
            for k, v in split(rows).items():
                if v: print(f'{counter:22s} {k[0]:18s} {k[1]:22s} {statistics.median(v[10:]):14.1f}')
try:
    print(open(raw + '/failed.txt').read())
except OSError:
    pass
PY
cat "$ROOT/gpurun_out/insitu_pmc_$TAG.txt" | head -80
