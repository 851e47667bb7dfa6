"""In-place forward (y == x, the operator's own contract): pattern-table kernel chunk length by tensor size, x freshly written
by a producer (a copy from a source tensor) + 100 MB of unrelated traffic before every launch -- the state inside a training
step -- and back to back."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store

dev = 'cuda'
other = torch.randn(50 * 2**20, device=dev).to(torch.bfloat16)
other2 = torch.empty_like(other)


def measure(kernel, before, reps=40):
    ts = []
    for i in range(reps + 8):
        before()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); kernel(); b.record()
        if i >= 8:
            ts.append((a, b))
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ts) * 1e3


for dtype, bits in ((torch.bfloat16, 3), (torch.float16, 4)):
    bo, lv = store.get('gelu' if bits == 3 else 'silu', bits, dev, dtype); bo = bo[1:-1].contiguous()
    fn = 'gelu' if bits == 3 else 'silu'
    for n in (4096 * 4096, 8192 * 4096, 16384 * 3072, 8192 * 8192, 2**27):
        src = torch.randn(n, device=dev).to(dtype)
        x = torch.empty_like(src)
        state = torch.empty(cabi.state_nbytes(n, bits), dtype=torch.uint8, device=dev)
        fb = n * (4 + bits / 8)

        def produced():
            x.copy_(src)
            other2.copy_(other)

        def inplace():
            cabi.quantize_forward(fn, x, bo, out=x, state=state)

        row = []
        for c in (-1, 0, 3, 4, 5, 6, 7, 8):
            cabi.tune(lut_chunk=c)
            plan = cabi.describe_forward(fn, dtype, n, bo.numel())
            t1 = measure(inplace, produced)
            x.copy_(src)
            t2 = measure(inplace, lambda: None)
            row.append(f'{c}:{plan["chunk"]} {t1:.1f}/{t2:.1f}')
        cabi.tune(lut_chunk=-1)
        print(f'{str(dtype)[6:]:9s} k={bits} n={n:10d}  lut_chunk:chunk  us produced/back-to-back   ' + '   '.join(row), flush=True)
