"""Random-projection product S.M on MI355X: the Philox-in-register MFMA kernel (fewbit_hip_sketch) against what it replaces
(draw S into HBM with torch.randn / randint, then torch.matmul = hipBLASLt), at the shapes of RoBERTa-base's linear layers
(rows = 128 x 128 tokens).  TFLOP/s = 2 * proj * rows * features / time; peak = 2500 (bf16 dense)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/ -> repository root
sys.path.insert(0, ROOT)
import torch
from fewbit_amd import cabi

DEV = 'cuda'
PEAK = 2500.0


SETTLE_S = 0.04


def timed(f, reps=100, warm=5, rounds=3, settle_s=SETTLE_S):
    """median over `rounds` of the average of `reps` back-to-back calls (HIP events on the launch stream), in us, after the GPU
    has been busy with the same call for `settle_s` (an idle GPU boosts, then dips for ~10 ms, then settles) -- the way bench.py
    and tools/profile_sketch.sh settle, so that the three agree"""
    for _ in range(warm):
        f()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        for _ in range(10):
            f()
        torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f()
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(out)[len(out) // 2]


def sampled_transforms(m, proj):
    """the reference's O(n log n) estimators at this shape (fewbit/functional/linear.py:113-131): time of dct(M, dim=0)[idx] and
    fft(M, dim=0)[idx] as fewbit_amd.linear runs them, beside the byte floor (read M once, write the sampled rows)"""
    from fewbit_amd import linear
    rows, features = m.shape
    gen = torch.Generator(device=DEV).manual_seed(3)
    floor_bytes = (rows + proj) * features * m.element_size()
    rec = {'byte_floor_bytes': floor_bytes, 'byte_floor_us_at_8TBs': round(floor_bytes / 8e6, 2)}
    for kind in ('dct', 'dft'):
        us = timed(lambda: linear.sampled_transform(kind, m, proj, gen), reps=20)
        rec[kind] = {'us': round(us, 1), 'x_byte_floor': round(us / (floor_bytes / 8e6), 1), 'path': linear.sampled_transform_path(kind, m)}
    return rec


def main():
    out = []
    slices = [int(s) for s in os.environ.get('SLICES', '-1').split(',')]
    shapes = [(16384, 768, 1638), (16384, 3072, 1638), (16384, 768, 3276), (16384, 3072, 3276), (16384, 3072, 8192), (65536, 4096, 4096)]
    if os.environ.get('QUICK'):
        shapes = shapes[:2]
    for dtype in (torch.bfloat16, torch.float32):
        for rows, features, proj in shapes:
            m = torch.randn(rows, features, device=DEV).to(dtype)
            flops = 2.0 * proj * rows * features
            rec = {'dtype': str(dtype).split('.')[-1], 'rows': rows, 'features': features, 'proj': proj}
            for dist in ('rademacher', 'gaussian'):
                for z in slices:
                    cabi.tune_sketch_slices(z)
                    plan = cabi.describe_sketch(dist, rows, features, proj, dtype)
                    ws = torch.empty(max(plan['workspace_bytes'], 1), dtype=torch.uint8, device=DEV)
                    o = torch.empty(proj, features, dtype=dtype, device=DEV)
                    us = timed(lambda: cabi.sketch(dist, m, proj, 1234, 1.0 / proj, out=o, workspace=ws))
                    rec[f'{dist}_z{plan["grid"][2]}'] = {'us': round(us, 1), 'TFLOPs': round(flops / us / 1e6, 1), 'frac_of_bf16_peak': round(flops / us / 1e6 / PEAK, 3),
                                                       'grid': plan['grid']}
            cabi.tune_sketch_slices(-1)
            # what it replaces: S in HBM + library GEMM (S drawn in the operand dtype of the matrix pipe)
            # (fp32 input: the cheapest torch formulation -- M rounded to bf16 INSIDE the timed call, as the kernel's own time
            # includes its conversion pass; the reference's own arithmetic, fp32 S and an fp32 GEMM, is timed beside it)
            op = torch.bfloat16 if dtype != torch.float16 else torch.float16
            mo = m.to(op)

            def torch_gauss():
                S = torch.randn(proj, rows, device=DEV, dtype=op)
                return (S @ (m.to(op) if dtype == torch.float32 else mo)) * (1.0 / proj)

            def torch_rad():
                S = torch.randint(0, 2, (proj, rows), device=DEV, dtype=torch.int8).to(op) * 2 - 1
                return (S @ (m.to(op) if dtype == torch.float32 else mo)) * (1.0 / proj)

            if dtype == torch.float32:
                rec['torch_fp32_randn_plus_fp32_matmul_us'] = round(timed(lambda: (torch.randn(proj, rows, device=DEV) @ m) * (1.0 / proj), reps=5, warm=2), 1)

            S = torch.randn(proj, rows, device=DEV, dtype=op)
            rec['torch_randn_plus_matmul_us'] = round(timed(torch_gauss), 1)
            rec['torch_randint_plus_matmul_us'] = round(timed(torch_rad), 1)
            rec['torch_matmul_only_us'] = round(timed(lambda: S @ mo), 1)
            rec['torch_matmul_only_TFLOPs'] = round(flops / rec['torch_matmul_only_us'] / 1e6, 1)
            rec['s_fragment_bytes'] = {dist: cabi.describe_sketch(dist, rows, features, proj, dtype)['s_fragment_bytes'] for dist in ('rademacher', 'gaussian')}
            if rows * features <= 16384 * 3072:
                rec['sampled_transform'] = sampled_transforms(m, proj)
            print(json.dumps(rec), flush=True)
            out.append(rec)
            del m, S, mo
            torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'sketch_bench.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
