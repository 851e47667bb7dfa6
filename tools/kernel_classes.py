#!/usr/bin/env python3
"""Group a rocprofv3 `--kernel-trace --stats` kernel_stats.csv into a few classes (where a training step's GPU time goes).

    python3 tools/kernel_classes.py <dir with *kernel_stats.csv> <steps the run made, warm-up included> [label]
prints one JSON object: GPU ms per step per class, the estimator kernels (dense sketches, sampled DCT) listed one by one.
"""
import csv
import glob
import json
import os
import sys

CLASSES = (('sketch', ('fewbit_hip::sketch::', 'fewbit_hip::dct::')), ('fewbit activation', ('fewbit_hip::', )),
           ('gemm', ('Cijk_', 'gemm', 'Gemm')), ('attention / softmax', ('softmax', 'Softmax', 'attn', 'fmha')),
           ('layer norm', ('layer_norm', 'LayerNorm', 'layernorm')), ('rng', ('philox', 'distribution', 'random')),
           ('copy / cast', ('copy', 'Copy', 'direct_copy', 'CatArray')), ('elementwise', ('elementwise', 'vectorized')),
           ('reduce', ('reduce', 'Reduce')))


def classify(name):
    return next((c for c, keys in CLASSES if any(k in name for k in keys)), 'other')


def main():
    root, steps = sys.argv[1], int(sys.argv[2])
    hits = sorted(glob.glob(os.path.join(root, '**', '*kernel_stats.csv'), recursive=True))
    if not hits:
        raise SystemExit(f'no kernel_stats.csv under {root}')
    acc, sketch, total = {}, [], 0.0
    with open(hits[0], newline='') as f:
        for row in csv.DictReader(f):
            name, ns, calls = row['Name'], float(row['TotalDurationNs']), int(row['Calls'])
            total += ns
            cls = classify(name)
            a = acc.setdefault(cls, [0.0, 0])
            a[0] += ns
            a[1] += calls
            if cls == 'sketch':
                sketch.append({'kernel': name.split('fewbit_hip::')[1].split('(')[0].replace('sketch::', ''), 'calls_per_step': round(calls / steps, 2),
                               'avg_us': round(ns / calls / 1e3, 1), 'ms_per_step': round(ns / steps / 1e6, 3)})
    out = {'label': sys.argv[3] if len(sys.argv) > 3 else root, 'steps': steps, 'gpu_ms_per_step': round(total / steps / 1e6, 3),
           'classes': {c: {'ms_per_step': round(v[0] / steps / 1e6, 3), 'launches_per_step': round(v[1] / steps, 1)}
                       for c, v in sorted(acc.items(), key=lambda kv: -kv[1][0])},
           'sketch_kernels': sorted(sketch, key=lambda r: -r['ms_per_step'])}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
