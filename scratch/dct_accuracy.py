"""Error of the sampled DCT against the float64 transform of the same data: max and rms of (got - want), relative to each element and to the column rms"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fewbit
from fewbit_amd import cabi
for rows, features, p, dtype, kind in ((16384, 768, 3276, torch.bfloat16, 'randn'), (16384, 768, 3276, torch.bfloat16, 'smooth'), (16384, 768, 3276, torch.bfloat16, 'tiny'), (16384, 768, 3276, torch.bfloat16, 'outlier'),
                                       (16384, 768, 3276, torch.float32, 'randn'), (4096, 96, 800, torch.bfloat16, 'randn')):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(rows, features, generator=g)
    if kind == 'smooth':
        x = torch.cumsum(x, 0) / 50
    if kind == 'tiny':
        x = x * 1e-7
    if kind == 'outlier':
        x[::997] *= 300
    x = x.to(dtype).cuda()
    idx = torch.randint(0, rows, (p, ), generator=g).cuda()
    got = cabi.sampled_dct(x, idx).double()
    want = fewbit.fft.dct(x.double(), dim=0, norm='ortho')[idx]
    err = (got - want).abs()
    rms = want.pow(2).mean().sqrt()
    print(f'{rows}x{features} p={p} {str(dtype)[6:]:9s} {kind:8s} rms(want) {float(rms):.3e}  rms(err)/rms {float(err.pow(2).mean().sqrt() / rms):.3e}  max(err)/rms {float(err.max() / rms):.3e}  '
          f'max(err/|want|) over |want|>rms/10: {float((err / want.abs())[want.abs() > rms / 10].max()):.3e}')
