#!/usr/bin/env python3
"""Summary of one tools/profile_dct.sh configuration: per kernel the average duration of the last `reps` dispatches of the rocprofv3
kernel trace, their sum beside the HIP-event time of the same calls, achieved GB/s of the bytes the design moves, and (PMC passes) the HBM
traffic per launch = 2 * FETCH_SIZE + WRITE_SIZE in KiB units with the gfx950 correction of MI355X_MICROARCH.md (as tools/summarize_profiles.py).

    python3 tools/dct_profile_summary.py <raw dir> <reps>
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
    raise SystemExit('tools/dct_run.py printed no line under the profiler')
print('# tools/profile_dct.sh: rocprofv3 --kernel-trace --stats -- python3 tools/dct_run.py', run['rows'], run['features'], run['proj'], run['dtype'], run['reps'], run['settle_ms'],
      {'fewbit_hip_sampled_dct_seeded': 'seeded', 'fewbit_hip_sampled_dct': 'explicit'}.get(run.get('what'), ''), f"   [{run.get('what')}]")
print(f"# {run['settle_calls']} settling calls precede the {run['reps']} timed calls; the table averages each kernel's last {reps} dispatches")
per = collections.defaultdict(list)
with open(find('trace', '*kernel_trace.csv'), newline='') as f:
    for row in csv.DictReader(f):
        if 'fewbit_hip::dct' in row['Kernel_Name']:
            per[row['Kernel_Name']].append((int(row['Start_Timestamp']), int(row['End_Timestamp'])))
es = 4 if run['dtype'] == 'f32' else 2
inter = ((run['features'] + 63) // 64) * run['rows'] * 256
alg = {'dct_pass_a_kernel': run['rows'] * run['features'] * es + inter, 'dct_pass_b_kernel': inter + run['proj'] * run['features'] * es}
total = 0.0
print(f"{'kernel':<52} {'calls':>6} {'avg us':>9} {'min us':>9} {'max us':>9} {'bytes by design':>16} {'GB/s':>8} {'of 8 TB/s':>10}")
for name, spans in sorted(per.items()):
    spans.sort()
    d = [(e - s) / 1e3 for s, e in spans[-reps:]]
    avg = sum(d) / len(d)
    total += avg
    short = name.replace('fewbit_hip::dct::', '').replace('void ', '')
    short = short[:short.index('(')] if '(' in short else short
    b = next((v for k, v in alg.items() if k in short), 0)
    print(f'{short:<52} {len(d):>6} {avg:>9.2f} {min(d):>9.2f} {max(d):>9.2f} {b:>16} {b / avg / 1e3:>8.1f} {b / avg / 1e3 / 8000:>10.4f}')
print(f'sum of the two kernels of one call (rocprofv3, settled): {total:.2f} us; HIP events around the same calls in the profiled process: {run["event_us_per_call"]:.2f} us')
print(f"byte floor (read M once, write the sampled rows) {run['byte_floor']} B = {run['byte_floor'] / 8e6:.2f} us at 8 TB/s -> the call is {total / (run['byte_floor'] / 8e6):.1f} x its floor; "
      f"bytes the design moves {run['bytes_moved_by_design']} B -> {run['bytes_moved_by_design'] / total / 1e3:.0f} GB/s = {run['bytes_moved_by_design'] / total / 1e3 / 8000:.3f} of 8 TB/s")
traffic = collections.defaultdict(dict)
for sub, counter in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    f = find(sub, '*counter_collection.csv')
    if not f:
        continue
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if 'fewbit_hip::dct' in r['Kernel_Name'] and r['Counter_Name'] == counter:
            k = 'pass_a' if 'pass_a' in r['Kernel_Name'] else 'pass_b'
            acc[k] += float(r['Counter_Value'])
            n[k].add(r['Dispatch_Id'])
    for k in acc:
        traffic[k][counter] = acc[k] / max(len(n[k]), 1)
if traffic:
    print('\n# HBM traffic per launch from the PMC passes (KiB counters; FETCH_SIZE x 2 on gfx950, as tools/summarize_profiles.py):')
    for k, d in sorted(traffic.items()):
        fetch, write = d.get('FETCH_SIZE', 0.0) * 2 * 1024, d.get('WRITE_SIZE', 0.0) * 1024
        b = alg['dct_pass_a_kernel' if k == 'pass_a' else 'dct_pass_b_kernel']
        print(f'  {k}: fetch {fetch / 1e6:.1f} MB + write {write / 1e6:.1f} MB = {(fetch + write) / 1e6:.1f} MB per launch = {(fetch + write) / b:.3f} x the bytes the design moves ({b / 1e6:.1f} MB)')
