"""First GPU contact: parity of the C-ABI against the oracle + crude timing."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import oracle
from fewbit_amd import cabi
from tests.helpers import forward_value_ok, assert_bit_equal

z = np.load('tests/golden/quantize_ref.npz')
dev = 'cuda'
tabs = {}
def table(name, k, dtype):
    tag = {torch.float32: 'f32', torch.bfloat16: 'bf16', torch.float16: 'f16'}[dtype]
    from tests.helpers import from_raw
    return from_raw(z[f'{name}{k:02d}_{tag}_borders'], dtype), from_raw(z[f'{name}{k:02d}_{tag}_levels'], dtype)

ok = True
for dtype in (torch.float32, torch.bfloat16, torch.float16):
    for k in (2, 3, 4):
        b, l = table('gelu', k, dtype)
        for n in (1, 7, 8, 9, 511, 512, 2048, 2049, 8191, 100003, 1 << 20):
            g = torch.Generator().manual_seed(n + k)
            x = (torch.randn(n, generator=g) * 1.5).to(dtype)
            gy = torch.randn(n, generator=g).to(dtype)
            if n > 64:
                x[:4] = torch.tensor([float('nan'), float('inf'), -float('inf'), -0.0]).to(dtype)
                x[4:4 + b.numel()] = b
            y_o, s_o, _ = oracle.quantize('gelu', x, b)
            gx_o = oracle.quantize_backward(gy, s_o, l)
            y_d, s_d = cabi.quantize_forward('gelu', x.to(dev), b.to(dev))
            gx_d = cabi.quantize_backward(gy.to(dev), s_d, l.to(dev))
            torch.cuda.synchronize()
            try:
                assert_bit_equal(s_d.cpu(), s_o, f'state {dtype} k{k} n{n}')
                assert_bit_equal(gx_d.cpu(), gx_o, f'gx {dtype} k{k} n{n}')
                fo = forward_value_ok(x, y_d.cpu(), y_o)
                assert fo.all(), (x[~fo][:4], y_d.cpu()[~fo][:4], y_o[~fo][:4])
            except AssertionError as e:
                ok = False
                print('FAIL', e)
print('parity', 'OK' if ok else 'FAILED')

# timing: 4096x4096 bf16 gelu k=3
n = 4096 * 4096
for dtype, k in ((torch.bfloat16, 3), (torch.float16, 2), (torch.float16, 4), (torch.float32, 3)):
    b, l = table('gelu', k, dtype)
    b, l = b.to(dev), l.to(dev)
    x = torch.randn(n, device=dev).to(dtype); gy = torch.randn(n, device=dev).to(dtype)
    y = torch.empty_like(x); gx = torch.empty_like(x)
    st = torch.empty(cabi.state_nbytes(n, k), dtype=torch.uint8, device=dev)
    for _ in range(20):
        cabi.quantize_forward('gelu', x, b, out=y, state=st); cabi.quantize_backward(gy, st, l, out=gx)
    torch.cuda.synchronize()
    for name, fn in (('fwd', lambda: cabi.quantize_forward('gelu', x, b, out=y, state=st)),
                     ('bwd', lambda: cabi.quantize_backward(gy, st, l, out=gx)),
                     ('copy', lambda: y.copy_(x))):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        iters = 200
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / iters
        es = x.element_size()
        nbytes = n * (2 * es + (k / 8 if name != 'copy' else 0))
        print(f'{dtype} k={k} {name}: {us:.2f} us  {nbytes / us / 1e6:.2f} TB/s')
