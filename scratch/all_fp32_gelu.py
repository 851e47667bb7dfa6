"""fp32 GELU forward for EVERY finite fp32 x >= 0 (2^31 - 2^23 patterns) against the float64 formula rounded to fp32 (torch's
double-precision erf on the GPU): histogram of ULP distances.  And for every finite x < 0: |dy| <= max(1 ULP, 2^-24 |x|)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store
dev = 'cuda'
b, _ = store.get('gelu', 3, dev, torch.float32); inner = b[1:-1].contiguous()
CH = 1 << 27
hist = torch.zeros(16, dtype=torch.int64, device=dev)
neg_bad = 0; neg_n = 0; worst_neg = 0.0
t0 = time.time()
for c in range(32):                                  # 32 x 2^27 = all 2^32 patterns
    bits = torch.arange(c * CH, (c + 1) * CH, device=dev, dtype=torch.int64).to(torch.int32)
    x = bits.view(torch.float32)
    y, _ = cabi.quantize_forward('gelu', x, inner)
    xd = x.double()
    exact = (xd * 0.5 * (1.0 + torch.erf(xd * 0.7071067811865476))).float()
    fin = torch.isfinite(x)
    pos = fin & (bits >= 0)
    if bool(pos.any()):
        d = (y.view(torch.int32)[pos].long() - exact.view(torch.int32)[pos].long()).abs().clamp(max=15)
        hist += torch.bincount(d, minlength=16)
    neg = fin & (bits < 0)
    if bool(neg.any()):
        err = (y[neg].double() - exact[neg].double()).abs()
        ulp = torch.maximum(exact[neg].double().abs() * 2.0**-23, torch.full_like(err, 2.0**-149))
        tol = torch.maximum(ulp, xd[neg].abs() * 2.0**-24)
        neg_bad += int((err > tol).sum()); neg_n += int(neg.sum())
        worst_neg = max(worst_neg, float((err / tol).max()))
    del bits, x, y, xd, exact
h = hist.tolist()
tot = sum(h)
print('finite x >= 0:', tot, 'inputs; ULP histogram vs correctly rounded float64 formula:', {i: v for i, v in enumerate(h) if v})
print('  within 1 ULP: %.6f %%, max ULP: %d' % (100.0 * (h[0] + h[1]) / tot, max(i for i, v in enumerate(h) if v)))
print('finite x < 0:', neg_n, 'inputs; outside max(1 ULP, 2^-24 |x|):', neg_bad, ' worst error / tolerance: %.3f' % worst_neg)
print('%.0f s' % (time.time() - t0))
