"""Every fp32 pattern through the precise-class forward of every continuous functor against ATen's own fp32 GPU kernels:
count of inputs where the result is more than 4 fp32 steps AND more than 1e-6 (absolute) away, NaN/inf disagreements."""
import os, sys
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
    viol = nanmis = infmis = 0; worst = (0.0, 0.0)
    for c in range(32):
        bits = torch.arange(c * CH, (c + 1) * CH, device=dev, dtype=torch.int64).to(torch.int32)
        x = bits.view(torch.float32)
        y, _ = cabi.quantize_forward(name, x, inner, *PAR.get(name, (0.0, 0.0)))
        e = ref(x)
        nanmis += int((torch.isnan(y) != torch.isnan(e)).sum())
        both = torch.isfinite(y) & torch.isfinite(e)
        infmis += int(((torch.isinf(y) != torch.isinf(e)) & ~torch.isnan(y) & ~torch.isnan(e)).sum())
        yi, ei = y.view(torch.int32).long(), e.view(torch.int32).long()
        yo = torch.where(yi < 0, -(yi & 0x7fffffff), yi); eo = torch.where(ei < 0, -(ei & 0x7fffffff), ei)
        steps = (yo - eo).abs()
        diff = (y.double() - e.double()).abs()
        bad = both & (steps > 4) & (diff > 1e-6)
        nb = int(bad.sum())
        if nb:
            i = torch.argmax(torch.where(bad, diff, torch.zeros_like(diff)))
            if float(diff[i]) > worst[1]: worst = (float(x[i]), float(diff[i]))
        viol += nb
        del bits, x, y, e, yi, ei, yo, eo, steps, diff, bad, both
    print(f'{name:11s}: >4 steps and >1e-6 away from ATen (GPU fp32): {viol}  NaN disagreements {nanmis}  inf disagreements {infmis}  worst (x, |dy|) {worst}', flush=True)
