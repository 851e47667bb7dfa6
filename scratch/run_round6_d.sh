#!/bin/bash
# round 6, call d: where the sampled-DCT kernel pair's time goes: variants of fewbit_dct.hip (idx prefetched before the FFT = base; 512 threads; phases compiled out)
# under rocprofv3 --kernel-trace --stats, per-kernel averages
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r06d_dct_variants_$(date +%H%M%S).txt; : > $OUT
[ -n "${PYTEST:-}" ] && { timeout 600 python3 -m pytest tests/test_gpu_dct.py -m gpu -q -x 2>&1 | tail -15 | cut -c1-300; }
for v in ${VARIANTS:-prod}; do
  for shape in "16384 768 3276 bf16" "16384 3072 3276 bf16" "16384 768 3276 f32"; do
    if [ $v = prod ]; then unset FEWBIT_HIP_LIB; else export FEWBIT_HIP_LIB=$PWD/scratch/libfewbit_hip_dct_$v.so; fi
    RAW=gpurun_out/prof_dctvar_$v; rm -rf $RAW
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o d -- python3 tools/dct_run.py $shape 100 30 ${MODE:-explicit} > $RAW.log 2>&1
    python3 - "$RAW" "$v" "$shape" >> $OUT <<'PY'
import csv, glob, sys
raw, v, shape = sys.argv[1:4]
f = glob.glob(raw + '/**/*kernel_stats.csv', recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if 'fewbit_hip::dct' in r['Name']] if f else []
d = {('A' if 'pass_a' in r['Name'] else 'B'): float(r['AverageNs']) / 1e3 for r in rows}
ev = [l for l in open(raw + '.log', errors='replace') if l.startswith('{')]
import json
e = json.loads(ev[-1])['event_us_per_call'] if ev else None
print(f"{v:<14} {shape:<22} pass A {d.get('A', 0):7.2f} us   pass B {d.get('B', 0):7.2f} us   sum {sum(d.values()):7.2f}   events {e}")
PY
  done
done
cat $OUT
