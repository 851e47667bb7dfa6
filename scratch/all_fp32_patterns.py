"""Every one of the 2^32 fp32 bit patterns through the fp32 forward (register search, split layout) and the backward:
codes against an independent formulation on the GPU (count of borders below x, NaN -> last code), gradients against
levels[code] * gy.  A one-off soak run (about 70 GiB of device memory traffic per table)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store
dev = 'cuda'
CH = 1 << 28                                      # patterns per chunk
tables = []
for k in (1, 2, 3, 4):
    b, l = store.get('gelu', k, dev, torch.float32)
    tables.append((f'gelu k={k}', b[1:-1].contiguous(), l))
edge = torch.tensor([-float('inf'), -1.0, -1e-45, -0.0, 1e-45, 1.0, float('inf')], device=dev)        # borders on special values
tables.append(('edge borders (+-inf, +-denormal, -0)', edge, torch.arange(8.0, device=dev)))
t0 = time.time()
# round 3: the launch policy picks the tile width by size (these 2^28-element chunks: forward U = 2, backward one tile per
# wave); every width the build holds is swept explicitly as well
for setting in ({}, {'u_fwd': 1, 'u_bwd': 2, 'chunk': 0}, {'u_fwd': 2, 'u_bwd': 1, 'chunk': 3}):
  cabi.tune(u_fwd=-1, u_bwd=-1, chunk=-1)
  cabi.tune(**setting)
  print('launch setting:', setting or 'built-in policy', cabi.describe_forward('identity', torch.float32, CH, 7)['kernel'],
        cabi.describe_backward(torch.float32, CH, 8)['kernel'], flush=True)
  for name, inner, levels in tables:
      k = cabi.bitwidth(levels.numel())
      bad = 0
      for c in range(1 << 32 >> 28):
          bits = torch.arange(c * CH, (c + 1) * CH, device=dev, dtype=torch.int64).to(torch.int32)      # wraps into the negative half
          x = bits.view(torch.float32)
          y, st = cabi.quantize_forward('identity', x, inner)
          codes = cabi.unpack_codes(st, CH, k)
          want = torch.zeros(CH, dtype=torch.int32, device=dev)
          for j in range(inner.numel()):
              want += (inner[j] < x).to(torch.int32)
          want = torch.where(torch.isnan(x), torch.full_like(want, inner.numel()), want)
          bad += int((codes != want).sum())
          assert torch.equal(y.view(torch.int32), bits)
          gy = torch.full((CH,), 1.5, device=dev)
          gx = cabi.quantize_backward(gy, st, levels)
          bad += int((gx != levels[want.long()] * 1.5).sum())
          del bits, x, y, st, codes, want, gy, gx
      print(f'{name}: all 2^32 fp32 patterns, mismatches = {bad}  ({time.time() - t0:.0f} s)', flush=True)
      assert bad == 0
