"""Every remainder: n = base + r for r in 0..1030 at sizes just above one resident generation (so the built-in policy and
the tails of both launch shapes are exercised), bf16 (pattern table) and fp32 (split layout), forward + backward against
an independent torch formulation on the GPU; guard bytes behind every output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(3)
bad = 0; cases = 0
for dtype, base in ((torch.bfloat16, 6 * 1024 * 1024), (torch.bfloat16, 17 * 1024 * 1024 + 512 * 3), (torch.float32, 4 * 1024 * 1024 + 512),
                    (torch.float32, 13 * 1024 * 1024), (torch.float16, 2 * 1024 * 1024)):
    b, l = store.get('gelu', 3, dev, dtype); inner = b[1:-1].contiguous()
    nmax = base + 1031
    X = (torch.randn(nmax, generator=g, device=dev) * 1.5).to(dtype)
    GY = torch.randn(nmax, generator=g, device=dev).to(dtype)
    codes_all = torch.bucketize(X.float(), inner.float(), out_int32=True)
    want_gx_all = (l.float()[codes_all.long()] * GY.float()).to(dtype)
    for r in range(0, 1031):
        n = base + r
        nbytes = cabi.state_nbytes(n, 3)
        y = torch.full((n + 16,), 7.0, device=dev, dtype=dtype); st = torch.full((nbytes + 16,), 0xAB, dtype=torch.uint8, device=dev)
        gx = torch.full((n + 16,), 7.0, device=dev, dtype=dtype)
        cabi.quantize_forward('gelu', X[:n], inner, out=y[:n], state=st[:nbytes])
        cabi.quantize_backward(GY[:n], st[:nbytes], l, out=gx[:n])
        ok = torch.equal(st[:nbytes], cabi.pack_codes(codes_all[:n], 3)) and bool((st[nbytes:] == 0xAB).all())
        ok = ok and torch.equal(gx[:n].view(torch.uint8), want_gx_all[:n].view(torch.uint8)) and bool((gx[n:] == 7.0).all()) and bool((y[n:] == 7.0).all())
        cases += 1
        if not ok:
            bad += 1
            print('MISMATCH', dtype, n)
    print(f'{str(dtype)[6:]} base {base}: 1031 sizes done, mismatches so far {bad}', flush=True)
print('cases', cases, 'mismatches', bad)
