#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh into the small summaries kept under profiles/.

  <round>_bench_kernel_stats.csv     rocprofv3 --kernel-trace --stats of `bench.py --no-extras` (the timed region's command)
  <round>_pmc_traffic.json           C2: HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KiB), gfx950 correction
  traffic_forward.json               the one number bench.py replays into roofline.traffic
  <round>_configs.json               every BASELINE config, warm and cold: rocprofv3 per-dispatch average duration of the
                                     forward and backward kernels, achieved GB/s, fraction of 8 TB/s, PMC traffic per launch
"""
import csv
import glob
import json
import os
import shutil
import sys

HBM_PEAK = 8000.0


def find(raw, sub, pattern):
    hits = sorted(glob.glob(os.path.join(raw, sub, '**', pattern), recursive=True))
    if not hits:
        raise SystemExit(f'no {pattern} under {raw}/{sub}')
    return hits[0]


def counter(raw, sub, name):
    """per-dispatch rows of one counter for the fewbit kernels, in dispatch order -> [(kernel name, value)]"""
    rows = []
    with open(find(raw, sub, '*counter_collection.csv'), newline='') as f:
        for row in csv.DictReader(f):
            if 'fewbit_hip::' in row['Kernel_Name'] and row['Counter_Name'] == name:
                rows.append((int(row['Dispatch_Id']), row['Kernel_Name'], float(row['Counter_Value'])))
    rows.sort()
    return [(k, v) for _, k, v in rows]


def trace(raw, sub):
    """fewbit kernel dispatches in start order -> [(kernel name, duration ns)]"""
    rows = []
    with open(find(raw, sub, '*kernel_trace.csv'), newline='') as f:
        for row in csv.DictReader(f):
            if 'fewbit_hip::' in row['Kernel_Name']:
                rows.append((int(row['Start_Timestamp']), row['Kernel_Name'], int(row['End_Timestamp']) - int(row['Start_Timestamp'])))
    rows.sort()
    return [(k, d) for _, k, d in rows]


def split(rows, phases):
    """cut the dispatch list into the phases of tools/config_runs.py"""
    need = sum(p['launches'] for p in phases)
    if len(rows) != need:
        raise SystemExit(f'{len(rows)} fewbit dispatches in the trace, {need} launches in phases.json')
    out, pos = [], 0
    for p in phases:
        out.append(rows[pos:pos + p['launches']])
        pos += p['launches']
    return out


def short(kernel):
    return kernel.split('fewbit_hip::')[1].split('(')[0]


def c2_traffic(rnd, raw, out):
    acc = {}
    for sub, name in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
        for kn, v in counter(raw, sub, name):
            acc.setdefault(('forward' if 'forward' in kn else 'backward', name), []).append(v)
    algorithmic = 4096 * 4096 * (2 * 2 + 3 / 8)
    kernels = {}
    for k in ('forward', 'backward'):
        f, w = acc[(k, 'FETCH_SIZE')], acc[(k, 'WRITE_SIZE')]
        fb, wb = int(round(sum(f) / len(f) * 1024 * 2)), int(round(sum(w) / len(w) * 1024))
        kernels[k] = {'FETCH_SIZE_KB_raw': round(sum(f) / len(f), 1), 'WRITE_SIZE_KB': round(sum(w) / len(w), 1),
                      'fetch_bytes_corrected': fb, 'write_bytes': wb, 'hbm_bytes_per_launch': fb + wb, 'dispatches': len(f),
                      'algorithmic_bytes_per_launch': int(algorithmic), 'traffic_over_algorithmic': round((fb + wb) / algorithmic, 4)}
    doc = {'round': int(rnd.lstrip('r')),
           'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py '
                      '--steps 50 --warmup 5 --no-extras --no-cpu-baseline (two separate passes, tools/profile_round.sh)',
           'correction': 'gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read (MI355X_MICROARCH.md, HBM section) -> '
                         'doubled; WRITE_SIZE as is; both in KiB',
           'workload': 'gelu bits=3, 4096x4096 bf16, one buffer set (cache-warm)', 'kernels': kernels}
    json.dump(doc, open(os.path.join(out, f'{rnd}_pmc_traffic.json'), 'w'), indent=1)
    json.dump({'hbm_bytes_per_launch': kernels['forward']['hbm_bytes_per_launch'], 'source': f'profiles/{rnd}_pmc_traffic.json'},
              open(os.path.join(out, 'traffic_forward.json'), 'w'), indent=1)
    return kernels


def configs(rnd, raw, out):
    ph = json.load(open(os.path.join(raw, 'cfg_trace', 'phases.json')))['phases']
    dur = split(trace(raw, 'cfg_trace'), ph)
    pmc = {}
    for sub, name in (('cfg_fetch', 'FETCH_SIZE'), ('cfg_write', 'WRITE_SIZE')):
        p2 = json.load(open(os.path.join(raw, sub, 'phases.json')))['phases']
        pmc[name] = (p2, split(counter(raw, sub, name), p2))
    doc = {'round': int(rnd.lstrip('r')),
           'command': 'rocprofv3 --kernel-trace | --pmc FETCH_SIZE | --pmc WRITE_SIZE (three separate passes) -- python3 '
                      'tools/config_runs.py; per-dispatch rows split into phases by launch count',
           'peak_GBps': HBM_PEAK, 'fetch_correction': 'FETCH_SIZE doubled (gfx950, wide coalesced reads)', 'configs': {}}
    for i, p in enumerate(ph):
        if p['mode'] == 'touch':
            continue
        rows = dur[i]
        entry = {'workload': p['workload'], 'buffer_sets': p['buffer_sets'], 'launches': p['launches'],
                 'algorithmic_bytes_per_launch': p['algorithmic_bytes_per_launch']}
        for which, sel in (('forward', rows[0::2]), ('backward', rows[1::2])):
            avg = sum(d for _, d in sel) / len(sel)
            entry[which] = {'kernel': short(sel[0][0]), 'avg_us': round(avg / 1e3, 3), 'min_us': round(min(d for _, d in sel) / 1e3, 3),
                            'GBps': round(p['algorithmic_bytes_per_launch'] / avg, 1),
                            'frac_of_peak': round(p['algorithmic_bytes_per_launch'] / avg / HBM_PEAK, 4)}
        for which, off in (('forward', 0), ('backward', 1)):
            f = [v for _, v in pmc['FETCH_SIZE'][1][i][off::2]]
            w = [v for _, v in pmc['WRITE_SIZE'][1][i][off::2]]
            hb = (sum(f) / len(f) * 2 + sum(w) / len(w)) * 1024
            entry[which]['hbm_bytes_per_launch'] = int(round(hb))
            entry[which]['traffic_over_algorithmic'] = round(hb / p['algorithmic_bytes_per_launch'], 4)
        fwd, bwd = entry['forward'], entry['backward']
        entry['fwd_plus_bwd'] = {'sum_avg_us': round(fwd['avg_us'] + bwd['avg_us'], 3),
                                 'frac_of_peak': round(2 * p['algorithmic_bytes_per_launch'] / ((fwd['avg_us'] + bwd['avg_us']) * 1e3) / HBM_PEAK, 4)}
        doc['configs'].setdefault(p['config'], {})[p['mode']] = entry
    json.dump(doc, open(os.path.join(out, f'{rnd}_configs.json'), 'w'), indent=1)
    return doc


def insitu(rnd, raw, out):
    """The few-bit kernels INSIDE a training step (tools/roberta_bench.py under rocprofv3 --kernel-trace --stats): x is fresh
    from the GEMM that produced it, gy from the GEMM's backward -- neither the cache-warm nor the cache-cold loop of bench.py.
    Activations of RoBERTa-base at batch 128 x seq 128: 16384 x 3072 elements per layer, 3 bits."""
    n = 16384 * 3072
    doc = {'round': int(rnd.lstrip('r')), 'elements_per_launch': n, 'peak_GBps': HBM_PEAK,
           'command': 'rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/roberta_bench.py --dtype <dt> --only fewbit|op '
                      '--steps 6 (tools/profile_round.sh); every dispatch of the run, warm-up steps included', 'runs': {}}
    for sub in sorted(glob.glob(os.path.join(raw, 'insitu_*'))):
        tag = os.path.basename(sub)[len('insitu_'):]
        es = 4 if 'fp32' in tag else 2
        nbytes = n * (2 * es + 3 / 8)
        try:
            stats = find(raw, os.path.basename(sub), '*kernel_stats.csv')
        except SystemExit:
            continue
        shutil.copy(stats, os.path.join(out, f'{rnd}_roberta_kernel_stats_{tag}.csv'))
        rows = {}
        with open(stats, newline='') as f:
            for row in csv.DictReader(f):
                if 'fewbit_hip::' in row['Name']:
                    avg = float(row['AverageNs'])
                    rows[short(row['Name'])] = {'calls': int(row['Calls']), 'avg_us': round(avg / 1e3, 2), 'min_us': round(float(row['MinNs']) / 1e3, 2),
                                                'max_us': round(float(row['MaxNs']) / 1e3, 2), 'pct_of_gpu_time': float(row['Percentage']),
                                                'algorithmic_bytes_per_launch': int(nbytes), 'GBps': round(nbytes / avg, 1),
                                                'frac_of_peak': round(nbytes / avg / HBM_PEAK, 4)}
        doc['runs'][tag] = rows
    if doc['runs']:
        json.dump(doc, open(os.path.join(out, f'{rnd}_roberta_insitu.json'), 'w'), indent=1)
    return doc


def main():
    rnd, raw, out = sys.argv[1:4]
    os.makedirs(out, exist_ok=True)
    if len(sys.argv) > 4 and sys.argv[4] == 'insitu':
        print(json.dumps(insitu(rnd, raw, out)['runs'], indent=1))
        return
    shutil.copy(find(raw, 'trace', '*kernel_stats.csv'), os.path.join(out, f'{rnd}_bench_kernel_stats.csv'))
    print(json.dumps(c2_traffic(rnd, raw, out), indent=1))
    doc = configs(rnd, raw, out)
    for name, modes in doc['configs'].items():
        for mode, e in modes.items():
            print(f"{name:8s} {mode:4s} fwd {e['forward']['avg_us']:8.2f} us {e['forward']['frac_of_peak']:.3f} (traffic x{e['forward']['traffic_over_algorithmic']:.3f})  "
                  f"bwd {e['backward']['avg_us']:8.2f} us {e['backward']['frac_of_peak']:.3f} (x{e['backward']['traffic_over_algorithmic']:.3f})  "
                  f"fwd+bwd {e['fwd_plus_bwd']['frac_of_peak']:.3f}")


if __name__ == '__main__':
    main()
