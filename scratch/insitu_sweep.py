"""Launch shapes of the bf16 forward IN PLACE on 16384 x 3072 right after its producing GEMM (+ 100 MB of other traffic): the
state the kernel meets inside a training step.  Tune keys of the production library (fewbit_hip_tune)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store

dev = 'cuda'
rows, din, dout = 16384, 768, 3072
bo, lv = store.get('gelu', 3, dev, torch.bfloat16); bo = bo[1:-1].contiguous()
h = torch.randn(rows, din, device=dev).to(torch.bfloat16)
w = (torch.randn(dout, din, device=dev) * 0.05).to(torch.bfloat16)
x = torch.empty(rows, dout, device=dev, dtype=torch.bfloat16)
y = torch.empty_like(x)
state = torch.empty(cabi.state_nbytes(x.numel(), 3), dtype=torch.uint8, device=dev)
other = torch.randn(50 * 2**20, device=dev).to(torch.bfloat16)
other2 = torch.empty_like(other)
fb = x.numel() * (2 * 2 + 3 / 8)


def before():
    torch.matmul(h, w.t(), out=x)
    other2.copy_(other)


def measure(kernel, reps=50):
    ts = []
    for i in range(reps + 8):
        before()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); kernel(); b.record()
        if i >= 8:
            ts.append((a, b))
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ts) * 1e3


def inplace():
    cabi.quantize_forward('gelu', x, bo, out=x, state=state)


def outofplace():
    cabi.quantize_forward('gelu', x, bo, out=y, state=state)


base = dict(lut_chunk=-1, lut_blocks_per_cu=-1, lut_min=-1, lut_block=-1, chunk=-1, waves_per_cu=-1, u_lut=-1, u_fwd=-1)
cases = [dict()] + [dict(lut_chunk=c) for c in (0, 1, 2, 3, 4, 6, 8, 12, 16, 32)] + [dict(lut_blocks_per_cu=1), dict(lut_blocks_per_cu=1, lut_chunk=1), dict(lut_blocks_per_cu=1, lut_chunk=4)] \
    + [dict(lut_min=1 << 40)] + [dict(lut_min=1 << 40, chunk=c) for c in (0, 1, 2, 4, 8)] + [dict(lut_min=1 << 40, waves_per_cu=wv) for wv in (16, 24, 32)]
for rnd in range(2):
    for c in cases:
        cabi.tune(**{**base, **c})
        plan = cabi.describe_forward('gelu', torch.bfloat16, x.numel(), 7)
        ti, to = measure(inplace), measure(outofplace)
        print(f'{str(c):50s} {plan["kernel"][:34]:34s} blocks {plan["blocks"]:6d} x {plan["threads"]:4d} chunk {plan["chunk"]:3d}   in place {ti:6.1f} us {fb / ti / 8e6:.3f}   y<-x {to:6.1f} us {fb / to / 8e6:.3f}', flush=True)
cabi.tune(**base)
