#!/bin/bash
# round 6, call i: time of the two DCT passes against the number of workgroups (features swept at 16384 rows): is the launch quantised in rounds of resident workgroups?
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r06i_dct_rounds.txt; : > $OUT
for f in ${FEATURES:-256 384 448 512 576 640 704 768 832 896 1024 1280 1536 2048 3072}; do
    RAW=gpurun_out/prof_dctrounds_$f; rm -rf $RAW
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o d -- python3 tools/dct_run.py 16384 $f 3276 bf16 100 30 explicit > $RAW.log 2>&1
    python3 - "$RAW" "$f" >> $OUT <<'PY'
import csv, glob, sys
raw, f = sys.argv[1], int(sys.argv[2])
g = glob.glob(raw + '/**/*kernel_stats.csv', recursive=True)
rows = [r for r in csv.DictReader(open(g[0])) if 'fewbit_hip::dct' in r['Name']] if g else []
d = {('A' if 'pass_a' in r['Name'] else 'B'): float(r['AverageNs']) / 1e3 for r in rows}
wa, wb = 128 * ((f + 63) // 64), 65 * ((f + 31) // 32)
print(f"features {f:5d}   pass A {wa:5d} workgroups = {wa / 1024:5.2f} rounds  {d.get('A', 0):7.2f} us = {d.get('A', 0) / (wa / 1024):6.2f} us per round"
      f"   pass B {wb:5d} workgroups = {wb / 1024:5.2f} rounds  {d.get('B', 0):7.2f} us = {d.get('B', 0) / (wb / 1024):6.2f} us per round")
PY
done
cat $OUT
