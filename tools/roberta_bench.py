#!/usr/bin/env python3
"""BASELINE.json configs[4]: RoBERTa-base forward+backward, all GELU -> fewbit.GELU(bits=3), batch 128 x seq 128:
peak memory and step time against the vanilla model.  Random-init weights of the roberta-base architecture
(`RobertaConfig()` defaults: 768 / 12 layers / 12 heads / 3072; no network access), synthetic token ids.

Two routes to the same kernels:
  --route module (default)  the 12 `intermediate_act_fn` modules are swapped for `fewbit.GELU(bits=k)` with `fewbit.map_module`;
  --route op                the REFERENCE'S OWN caller route: `intermediate_act_fn = lambda xs: torch.ops.fewbit.gelu(xs, bounds,
                            levels)` -- the raw operator with the literal 3-bit tables of benchmark/bench-roberta.py:128-137,
                            in place on the 3-D output of nn.Linear (a view of its 2-D addmm result), which is what the
                            reference's monkey patch of `transformers.activations.ACT2FN['gelu']` amounts to (:139-147);
  --route both              module and op side by side (adds `op` and `op_vs_module` to the line).
    python tools/roberta_bench.py [--dtype fp32|bf16] [--steps 10] [--bits 3] [--route module|op|both]
Prints one JSON line.
"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch  # noqa: E402

import fewbit  # noqa: E402


def build(dtype, device):
    from transformers import RobertaConfig, RobertaForSequenceClassification
    torch.manual_seed(0)
    cfg = RobertaConfig(num_labels=2)
    cfg.hidden_dropout_prob = 0.1
    model = RobertaForSequenceClassification(cfg)
    return model.to(device=device, dtype=dtype).train()


def swap_gelu(model, bits):
    from transformers.activations import GELUActivation
    n = [0]

    def fn(mod, path):
        if isinstance(mod, (GELUActivation, torch.nn.GELU)) and path.endswith('intermediate_act_fn'):
            n[0] += 1
            return fewbit.GELU(bits=bits)
        return mod

    fewbit.map_module(model, fn)
    return n[0]


# the literal tables of the reference's benchmark (benchmark/bench-roberta.py:128-137): 7 inner borders, 8 levels
REF_BOUNDS = (-2.39798704e+00, -7.11248159e-01, -3.26290283e-01, -1.55338428e-04, 3.26182064e-01, 7.10855860e-01, 2.39811567e+00)
REF_LEVELS = (-0.00260009, -0.08883533, 0.1251944, 0.37204148, 0.6277958, 0.87466175, 1.08880716, 1.00259936)


def patch_gelu_with_raw_op(model, dtype, device):
    """The reference's caller route: every intermediate activation becomes a plain function around the raw operator."""
    bounds = torch.tensor(REF_BOUNDS, device=device).to(dtype)
    levels = torch.tensor(REF_LEVELS, device=device).to(dtype)

    def gelu3bit(xs):
        return torch.ops.fewbit.gelu(xs, bounds, levels)

    n = 0
    for layer in model.roberta.encoder.layer:
        if 'intermediate_act_fn' in layer.intermediate._modules:       # registered as a sub-module: unregister first
            del layer.intermediate.intermediate_act_fn
        layer.intermediate.intermediate_act_fn = gelu3bit
        n += 1
    return n


def swap_linear(model, ratio, sketch_dtype=None, matmul='gaussian'):
    """Every nn.Linear of the encoder -> fewbit.RandomizedLinear(proj_dim_ratio=ratio) sharing its parameters (the
    README's `convert_linear` recipe); the classification head stays exact."""
    from fewbit.util import convert_linear
    n = [0]

    def fn(mod, path):
        if type(mod) is torch.nn.Linear and '/encoder/' in path:
            n[0] += 1
            return convert_linear(mod, fewbit.RandomizedLinear, proj_dim_ratio=ratio, sketch_dtype=sketch_dtype, matmul=matmul)
        return mod

    fewbit.map_module(model, fn)
    return n[0]


def run(model, ids, labels, steps, warmup=3):
    opt = torch.optim.SGD(model.parameters(), lr=1e-4)
    dev = ids.device

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(input_ids=ids, labels=labels)
        out.loss.backward()
        opt.step()
        return out.loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    base = torch.cuda.memory_allocated(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    peak = torch.cuda.max_memory_allocated(dev)
    with fewbit.memory_usage_hooks() as usage:
        model(input_ids=ids, labels=labels).loss.backward()
    return {'ms_per_step': dt * 1e3, 'peak_bytes': peak, 'peak_minus_resident_bytes': peak - base,
            'saved_for_backward_bytes': usage.forward, 'loss': float(loss)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='fp32', choices=('fp32', 'bf16'))
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--bits', type=int, default=3)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--seq', type=int, default=128)
    ap.add_argument('--only', default=None, choices=(None, 'vanilla', 'fewbit', 'op'), help='run a single variant (profiling)')
    ap.add_argument('--route', default='module', choices=('module', 'op', 'both'), help='see the module docstring')
    ap.add_argument('--table', action='store_true',
                    help="the four rows of the reference README's table: GELU {vanilla, 3-bit} x linear {vanilla, randomized}")
    ap.add_argument('--row', type=int, default=None, choices=(0, 1, 2, 3), help='with --table: run only this row (profiling)')
    ap.add_argument('--linear-ratio', type=float, default=0.2, help='proj_dim_ratio of the randomized linear layers')
    ap.add_argument('--sketch-bf16', action='store_true', help='run the sketch GEMMs of fp32 layers in bf16')
    ap.add_argument('--matmul', default='gaussian', choices=('gaussian', 'rademacher', 'dct', 'dft'),
                    help="kind of sketch of the randomized layers: the dense ones, or the reference's sampled transforms (fewbit/functional/linear.py:113-131)")
    ap.add_argument('--torch-sketch', action='store_true',
                    help='draw S with torch.randn / randint and multiply with torch.matmul (what round 3 measured) instead of the '
                         'gfx950 sketch kernels (fewbit_amd/csrc/fewbit_sketch.hip)')
    args = ap.parse_args()
    dtype = {'fp32': torch.float32, 'bf16': torch.bfloat16}[args.dtype]
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(5, 50000, (args.batch, args.seq), generator=g).to(dev)
    labels = torch.randint(0, 2, (args.batch,), generator=g).to(dev)

    import fewbit_amd.linear
    if args.torch_sketch:
        fewbit_amd.linear.use_native_sketch(False)
    if args.table:
        rows = []
        for i, (gelu, linear) in enumerate(((False, False), (True, False), (False, True), (True, True))):
            if args.row is not None and i != args.row:
                continue
            model = build(dtype, dev)
            ng = swap_gelu(model, args.bits) if gelu else 0
            nl = swap_linear(model, args.linear_ratio, torch.bfloat16 if args.sketch_bf16 else None, args.matmul) if linear else 0
            r = run(model, ids, labels, args.steps)
            rows.append({'gelu': f'{args.bits}-bit' if gelu else 'vanilla', 'linear': 'randomized' if linear else 'vanilla',
                         'gelu_modules_swapped': ng, 'linear_modules_swapped': nl, 'ms_per_step': round(r['ms_per_step'], 2),
                         'peak_gib': round(r['peak_bytes'] / 2**30, 3), 'loss': r['loss']})
            del model
            torch.cuda.empty_cache()
        for r in rows:
            r['saving_pct'] = round(100.0 * (1.0 - r['peak_gib'] / rows[0]['peak_gib']), 1)
            r['step_time_ratio'] = round(r['ms_per_step'] / rows[0]['ms_per_step'], 3)
        print(json.dumps({'config': f'RoBERTa-base (random init) batch {args.batch} x seq {args.seq}, {args.dtype}, '
                                    f'fwd+bwd+SGD step; randomized linear proj_dim_ratio={args.linear_ratio}'
                                    + f', {args.matmul} sketch, ' + (fewbit_amd.linear.sampled_transform_path(args.matmul, torch.empty(args.batch * args.seq, 8, device=dev, dtype=dtype)) if args.matmul in ('dct', 'dft') else
                                                                     'torch.randn/randint + torch.matmul' if args.torch_sketch else 'gfx950 sketch kernels (fewbit_hip_sketch)')
                                    + (', sketch GEMMs in bf16' if args.sketch_bf16 else ''),
                          'rows': rows}))
        return

    res = {}
    names = ('vanilla', 'fewbit') + (('op', ) if args.route in ('op', 'both') or args.only == 'op' else ())
    for name in names:
        if args.only and name != args.only:
            continue
        if args.route == 'op' and name == 'fewbit' and not args.only:
            continue
        model = build(dtype, dev)
        if name == 'op':
            swapped = patch_gelu_with_raw_op(model, dtype, dev)
        else:
            swapped = swap_gelu(model, args.bits) if name == 'fewbit' else 0
        res[name] = run(model, ids, labels, args.steps)
        res[name]['gelu_modules_swapped'] = swapped
        del model
        torch.cuda.empty_cache()
    if args.only:
        print(json.dumps({args.only: res[args.only]}))
        return
    es = 4 if dtype == torch.float32 else 2
    n_act = 12 * args.batch * args.seq * 3072
    expect = n_act * es - (args.bits * n_act) // 8
    out = {'config': f'RoBERTa-base (random init) batch {args.batch} x seq {args.seq}, {args.dtype}, fwd+bwd+SGD step',
           'bits': args.bits, 'vanilla': res['vanilla'], 'expected_saved_tensor_saving_bytes': expect}
    for key in ('fewbit', 'op'):
        if key not in res:
            continue
        out[key] = res[key]
        out[key + '_summary' if key == 'op' else 'summary'] = {
            'route': 'fewbit.GELU module (map_module)' if key == 'fewbit' else
                     'raw torch.ops.fewbit.gelu with the reference benchmark\'s literal tables, in place on the Linear output',
            'peak_saving_bytes': res['vanilla']['peak_bytes'] - res[key]['peak_bytes'],
            'saved_tensor_saving_bytes': res['vanilla']['saved_for_backward_bytes'] - res[key]['saved_for_backward_bytes'],
            'step_time_ratio': res[key]['ms_per_step'] / res['vanilla']['ms_per_step']}
    if 'fewbit' in res:       # (kept at top level: what earlier rounds' profiles/ files hold)
        out['peak_saving_bytes'] = out['summary']['peak_saving_bytes']
        out['saved_tensor_saving_bytes'] = out['summary']['saved_tensor_saving_bytes']
        out['step_time_ratio'] = out['summary']['step_time_ratio']
    if 'fewbit' in res and 'op' in res:
        out['op_vs_module'] = {'step_time_ratio': res['op']['ms_per_step'] / res['fewbit']['ms_per_step'],
                               'saved_bytes_difference': res['op']['saved_for_backward_bytes'] - res['fewbit']['saved_for_backward_bytes'],
                               'peak_bytes_difference': res['op']['peak_bytes'] - res['fewbit']['peak_bytes']}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
