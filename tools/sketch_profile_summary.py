#!/usr/bin/env python3
"""Summary of one tools/profile_sketch.sh configuration: per kernel the average duration of the LAST `reps` dispatches in the
rocprofv3 kernel trace (the settling dispatches before them are dropped), their sum per call beside the HIP-event time of the
same launches (measured inside the profiled process) -- and, when the PMC passes ran, the counters per kernel.

    python3 tools/sketch_profile_summary.py <raw dir> <reps>
"""
import collections
import csv
import glob
import json
import os
import sys

raw, reps = sys.argv[1], int(sys.argv[2])


def find(sub, pattern):
    hits = sorted(glob.glob(os.path.join(raw, sub, '**', pattern), recursive=True))
    return hits[0] if hits else None


run = None
for line in open(os.path.join(raw, 'run.log'), errors='replace'):
    if line.startswith('{'):
        run = json.loads(line)
if run is None:
    print(open(os.path.join(raw, 'run.log'), errors='replace').read()[-3000:])
    raise SystemExit('tools/sketch_run.py printed no line under the profiler')
print('# tools/profile_sketch.sh: rocprofv3 --kernel-trace --stats -- python3 tools/sketch_run.py', run['dist'], run['rows'], run['features'], run['proj'], run['dtype'],
      run['reps'], run['settle_ms'])
print('# plan:', json.dumps(run['plan']))
print(f"# {run['settle_calls']} settling calls ({run['settle_ms']} ms) precede the {run['reps']} timed calls; the table averages each kernel's last {reps} dispatches")

per = collections.defaultdict(list)
with open(find('trace', '*kernel_trace.csv'), newline='') as f:
    for row in csv.DictReader(f):
        if 'fewbit_hip::' in row['Kernel_Name']:
            per[row['Kernel_Name']].append((int(row['Start_Timestamp']), int(row['End_Timestamp'])))
flops = 2.0 * run['proj'] * run['rows'] * run['features']
total = 0.0
first_start, last_end = None, None
print(f"{'kernel':<70} {'calls':>6} {'avg us':>9} {'min us':>9} {'max us':>9} {'avg us (all dispatches)':>24}")
for name, spans in sorted(per.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    spans.sort()
    tail = spans[-reps:]
    d = [(e - s) / 1e3 for s, e in tail]
    d_all = [(e - s) / 1e3 for s, e in spans]
    total += sum(d) / len(d)
    first_start = tail[0][0] if first_start is None else min(first_start, tail[0][0])
    last_end = tail[-1][1] if last_end is None else max(last_end, tail[-1][1])
    short = name.replace('fewbit_hip::sketch::', '').replace('void ', '')
    short = short[:short.index('(')] if '(' in short else short
    print(f'{short:<70} {len(tail):>6} {sum(d) / len(d):>9.2f} {min(d):>9.2f} {max(d):>9.2f} {sum(d_all) / len(d_all):>24.2f}')
span_us = (last_end - first_start) / 1e3 / reps
print(f'sum of the kernels of one call (rocprofv3, settled): {total:.2f} us  ->  {flops / total / 1e6:.1f} TFLOP/s = {flops / total / 1e6 / 2500.0:.4f} of the 2.5 PFLOP/s dense bf16 peak')
print(f'first start to last end of the timed dispatches / {reps}: {span_us:.2f} us per call (kernels + the gaps between them, under the profiler)')
print(f"HIP events around the same {run['reps']} calls, inside the profiled process: {run['event_us_per_call']:.2f} us per call")
stats = find('trace', '*kernel_stats.csv')
if stats:
    print('\n# rocprofv3 --stats (ALL dispatches, the settling ones included):')
    for i, line in enumerate(open(stats)):
        if i == 0 or 'fewbit_hip::' in line:
            print(line.rstrip()[:260])
for sub in ('pmc_sq', 'pmc_tcc'):
    f = find(sub, '*counter_collection.csv')
    if not f:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if 'fewbit_hip::' not in r['Kernel_Name']:
            continue
        k = r['Kernel_Name'].replace('fewbit_hip::sketch::', '').replace('void ', '')
        k = k[:k.index('(')] if '(' in k else k
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        disp[k].add(r['Dispatch_Id'])
    print(f'\n# {sub} (its own pass, 20 calls, per dispatch):')
    for k, d in acc.items():
        n = max(len(disp[k]), 1)
        vals = {c: round(v / n) for c, v in sorted(d.items())}
        extra = ''
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and d.get('SQ_BUSY_CYCLES'):
            # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD... the ratio the round-5 verdict quotes: MFMA busy / (4 SIMDs x SQ_BUSY_CYCLES per CU) folded as in r05
            extra = f"  mfma_busy/busy = {d['SQ_VALU_MFMA_BUSY_CYCLES'] / d['SQ_BUSY_CYCLES'] / 32:.3f}"
        if d.get('TCC_REQ_sum'):
            extra = f"  L2 hit rate = {d['TCC_HIT_sum'] / d['TCC_REQ_sum']:.3f}"
        print(' ', k, vals, extra)
