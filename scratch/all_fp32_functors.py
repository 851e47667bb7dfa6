"""Every fp32 bit pattern through the precise-class forward of every continuous functor, against torch's float64
evaluation rounded to fp32: NaN mismatches, max / distribution of ULP distance over finite results (one-off soak)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
import torch.nn.functional as F
from fewbit_amd import cabi
dev = 'cuda'
inner = torch.tensor([-1.0, 0.0, 1.0], device=dev)
REF = {'celu': lambda x: F.celu(x, 1.3), 'elu': lambda x: F.elu(x, 0.7), 'gelu': F.gelu, 'hardswish': F.hardswish,
       'logsigmoid': F.logsigmoid, 'mish': F.mish, 'selu': F.selu, 'sigmoid': torch.sigmoid, 'silu': F.silu,
       'softplus': lambda x: F.softplus(x, 2.0, 5.0), 'softsign': F.softsign, 'tanh': torch.tanh, 'tanhshrink': F.tanhshrink}
PAR = {'celu': (1.3, 0.0), 'elu': (0.7, 0.0), 'softplus': (2.0, 5.0)}
CH = 1 << 27
for name, ref in REF.items():
    t0 = time.time()
    hist = torch.zeros(9, dtype=torch.int64, device=dev); nan_mismatch = 0; worst = 0
    for c in range(32):
        bits = torch.arange(c * CH, (c + 1) * CH, device=dev, dtype=torch.int64).to(torch.int32)
        x = bits.view(torch.float32)
        y, _ = cabi.quantize_forward(name, x, inner, *PAR.get(name, (0.0, 0.0)))
        e = ref(x.double()).float()
        nan_mismatch += int((torch.isnan(y) != torch.isnan(e)).sum())
        ok = torch.isfinite(y) & torch.isfinite(e)
        yi, ei = y.view(torch.int32).long(), e.view(torch.int32).long()
        yo = torch.where(yi < 0, -(yi & 0x7fffffff), yi); eo = torch.where(ei < 0, -(ei & 0x7fffffff), ei)
        d = (yo - eo).abs()[ok]
        worst = max(worst, int(d.max()))
        hist += torch.bincount(d.clamp(max=8), minlength=9)
        inf_mismatch = int(((torch.isinf(y) != torch.isinf(e)) & ~torch.isnan(y) & ~torch.isnan(e)).sum())
        nan_mismatch += 0
        del bits, x, y, e, yi, ei, yo, eo, d, ok
    h = hist.tolist()
    print(f'{name:11s}: NaN mismatches {nan_mismatch}, max ULP {worst}, ULP histogram (8 = 8 or more) {dict((i, v) for i, v in enumerate(h) if v)}  ({time.time()-t0:.0f} s)', flush=True)
