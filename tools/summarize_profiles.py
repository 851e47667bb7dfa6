#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh into the small summaries kept under profiles/.

  <round>_bench_kernel_stats.csv         rocprofv3 --kernel-trace --stats, kernel_stats table as emitted
  <round>_pmc_{fetch,write}_counter_collection.csv   per-dispatch counter rows of the two fewbit kernels only
  <round>_pmc_traffic.json               HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KiB), gfx950 correction
  traffic_forward.json                   the one number bench.py puts into roofline.traffic
"""
import csv
import glob
import json
import os
import shutil
import sys

ALGORITHMIC = 73400320          # 4096*4096 * (2*2 + 3/8) bytes per launch, forward and backward alike


def find(raw, sub, pattern):
    hits = sorted(glob.glob(os.path.join(raw, sub, '**', pattern), recursive=True))
    if not hits:
        raise SystemExit(f'no {pattern} under {raw}/{sub}')
    return hits[0]


def counter(raw, sub, name, out_csv):
    src = find(raw, sub, '*counter_collection.csv')
    acc = {'forward': [], 'backward': []}
    with open(src, newline='') as f, open(out_csv, 'w', newline='') as g:
        rd = csv.DictReader(f)
        wr = csv.DictWriter(g, fieldnames=rd.fieldnames, quoting=csv.QUOTE_NONNUMERIC)
        wr.writeheader()
        for row in rd:
            kn = row['Kernel_Name']
            if 'fewbit_hip::' not in kn or row['Counter_Name'] != name:
                continue
            wr.writerow(row)
            acc['forward' if 'forward' in kn else 'backward'].append(float(row['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items() if v}


def main():
    rnd, raw, out = sys.argv[1:4]
    os.makedirs(out, exist_ok=True)
    shutil.copy(find(raw, 'trace', '*kernel_stats.csv'), os.path.join(out, f'{rnd}_bench_kernel_stats.csv'))
    fetch = counter(raw, 'fetch', 'FETCH_SIZE', os.path.join(out, f'{rnd}_pmc_fetch_counter_collection.csv'))
    write = counter(raw, 'write', 'WRITE_SIZE', os.path.join(out, f'{rnd}_pmc_write_counter_collection.csv'))
    kernels = {}
    for k in ('forward', 'backward'):
        f_kb, nf = fetch[k]
        w_kb, _ = write[k]
        fb, wb = int(round(f_kb * 1024 * 2)), int(round(w_kb * 1024))
        kernels[k] = {'FETCH_SIZE_KB_raw': round(f_kb, 1), 'WRITE_SIZE_KB': round(w_kb, 1), 'fetch_bytes_corrected': fb,
                      'write_bytes': wb, 'hbm_bytes_per_launch': fb + wb, 'dispatches': nf,
                      'algorithmic_bytes_per_launch': ALGORITHMIC,
                      'traffic_over_algorithmic': round((fb + wb) / ALGORITHMIC, 4)}
    doc = {'round': int(rnd.lstrip('r')),
           'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py '
                      '--steps 50 --warmup 5 --no-cpu-baseline (two separate passes, tools/profile_round.sh)',
           'correction': 'gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read (MI355X_MICROARCH.md, HBM section) -> '
                         'doubled; WRITE_SIZE as is; both in KiB',
           'workload': 'gelu bits=3, 4096x4096 bf16', 'kernels': kernels}
    with open(os.path.join(out, f'{rnd}_pmc_traffic.json'), 'w') as f:
        json.dump(doc, f, indent=1)
    with open(os.path.join(out, 'traffic_forward.json'), 'w') as f:
        json.dump({'hbm_bytes_per_launch': kernels['forward']['hbm_bytes_per_launch'],
                   'source': f'profiles/{rnd}_pmc_traffic.json'}, f, indent=1)
    print(json.dumps(doc['kernels'], indent=1))


if __name__ == '__main__':
    main()
