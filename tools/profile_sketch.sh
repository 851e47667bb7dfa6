#!/bin/bash
# rocprofv3 passes over the random-projection kernel (scratch/sketch_run.py): kernel durations, then PMC counters in their own
# passes (never combined with other trace domains than --kernel-trace).  usage (through gpurun): bash tools/profile_sketch.sh <tag> [args of sketch_run.py]
set -u
TAG=${1:-r04}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; export TMPDIR=/tmp
RAW=$ROOT/gpurun_out/prof_sketch_$TAG; rm -rf "$RAW"; mkdir -p "$RAW"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/trace" -o sk -- python3 scratch/sketch_run.py "$@" > "$RAW/run.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$RAW/pmc_sq" -o sk -- python3 scratch/sketch_run.py "$@" > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d "$RAW/pmc_sq2" -o sk -- python3 scratch/sketch_run.py "$@" > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$RAW/pmc_fetch" -o sk -- python3 scratch/sketch_run.py "$@" > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d "$RAW/pmc_tcc" -o sk -- python3 scratch/sketch_run.py "$@" > /dev/null 2>&1
python3 - "$RAW" <<'PY'
import csv, glob, sys, collections
raw = sys.argv[1]
for f in glob.glob(raw + '/trace/**/*kernel_stats.csv', recursive=True):
    print(open(f).read()[:1500])
for sub in ('pmc_sq', 'pmc_sq2', 'pmc_fetch', 'pmc_tcc'):
    for f in glob.glob(f'{raw}/{sub}/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, d in acc.items():
            disp = len(set())
            print(sub, k, {c: v for c, v in d.items()})
PY
