"""More exhaustive sweeps (one-off): (1) the 1-bit family over all 2^32 fp32 inputs vs torch on the GPU, (2) the wide-table
fp32 forward (LDS tree search) over all 2^32 inputs, (3) the backward product for every (16-bit gy pattern, level) pair."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
import torch.nn.functional as F
from fewbit_amd import cabi
dev = 'cuda'
CH = 1 << 27
STEP = {'hardshrink': ((0.5,), lambda x: F.hardshrink(x, 0.5), lambda x: (x < -0.5) | (x > 0.5)),
        'hardsigmoid': ((), F.hardsigmoid, lambda x: ~((x <= -3) | (x >= 3))),
        'hardtanh': ((-1.0, 1.0), lambda x: F.hardtanh(x, -1.0, 1.0), lambda x: ~((x <= -1) | (x >= 1))),
        'leaky_relu': ((0.01,), lambda x: F.leaky_relu(x, 0.01), lambda x: ~(x >= 0)),
        'relu': ((), F.relu, lambda x: ~(x <= 0)),
        'relu6': ((), F.relu6, lambda x: ~((x <= 0) | (x >= 6))),
        'softshrink': ((0.5,), lambda x: F.softshrink(x, 0.5), lambda x: (x < -0.5) | (x > 0.5)),
        'threshold': ((0.25, -3.0), lambda x: F.threshold(x, 0.25, -3.0), lambda x: ~(x <= 0.25))}
for name, (p, ref, rule) in STEP.items():
    bad_y = bad_bits = 0
    for c in range(32):
        bits = torch.arange(c * CH, (c + 1) * CH, device=dev, dtype=torch.int64).to(torch.int32)
        x = bits.view(torch.float32)
        y, st = cabi.stepwise1_forward(name, x, *p)
        want = ref(x)
        neq = (y.view(torch.int32) != want.view(torch.int32)) & ~(torch.isnan(y) & torch.isnan(want))
        neq &= ~((y == 0) & (want == 0))                      # sign of a zero result: the kernels' rule gives +0, ATen keeps -0 in places
        bad_y += int(neq.sum())
        bad_bits += int((cabi.unpack_codes(st, CH, 1) != rule(x).to(torch.int32)).sum())
        del bits, x, y, st, want, neq
    print(f'1-bit {name:11s}: value mismatches {bad_y}, bit mismatches {bad_bits}', flush=True)
g = torch.Generator(device=dev).manual_seed(1)
for nlev in (33, 256):
    inner = torch.unique(torch.sort(torch.randn(nlev - 1, generator=g, device=dev) * 2)[0])
    k = cabi.bitwidth(inner.numel() + 1)
    bad = 0
    for c in range(32):
        bits = torch.arange(c * CH, (c + 1) * CH, device=dev, dtype=torch.int64).to(torch.int32)
        x = bits.view(torch.float32)
        _, st = cabi.quantize_forward('identity', x, inner)
        want = torch.bucketize(x, inner, out_int32=True)      # #{b < x}; NaN handled below
        want = torch.where(torch.isnan(x), torch.full_like(want, inner.numel()), want)
        bad += int((cabi.unpack_codes(st, CH, k) != want).sum())
        del bits, x, st, want
    print(f'wide fp32 forward, {inner.numel()} borders ({k} bits): code mismatches {bad}', flush=True)
for dtype in (torch.bfloat16, torch.float16):
    pat = torch.arange(65536, device=dev, dtype=torch.int32).to(torch.int16).view(dtype)
    levels = torch.cat([torch.tensor([0.0, -0.0, 1.0, -1.0, 0.5, 3.0, 1e-3, -2.5e-2], device=dev),
                        torch.randn(248, generator=g, device=dev)]).to(dtype)                         # 256 levels -> 8-bit codes
    codes = torch.arange(256, device=dev, dtype=torch.int32).repeat_interleave(65536)
    gy = pat.repeat(256)
    st = cabi.pack_codes(codes, 8)
    gx = cabi.quantize_backward(gy, st, levels)
    want = (levels.float()[codes.long()] * gy.float()).to(dtype)
    neq = (gx.view(torch.int16) != want.view(torch.int16)) & ~(torch.isnan(gx) & torch.isnan(want))
    print(f'backward product, every {str(dtype)[6:]} gy pattern x 256 levels: mismatches {int(neq.sum())}', flush=True)
    for kk, nl in ((3, 8), (1, 2)):
        lv = levels[:nl].contiguous()
        codes = torch.arange(nl, device=dev, dtype=torch.int32).repeat_interleave(65536)
        gy = pat.repeat(nl)
        gx = cabi.quantize_backward(gy, cabi.pack_codes(codes, kk), lv)
        want = (lv.float()[codes.long()] * gy.float()).to(dtype)
        neq = (gx.view(torch.int16) != want.view(torch.int16)) & ~(torch.isnan(gx) & torch.isnan(want))
        print(f'   ... x {nl} levels ({kk}-bit kernel): mismatches {int(neq.sum())}', flush=True)
