"""Why is the bf16 forward at 0.67-0.68 of the roofline inside the RoBERTa step and 0.72-0.77 stand-alone at similar sizes?
The kernel (in place on a 16384 x 3072 bf16 tensor, as in the model) and a plain copy of the same tensor, each timed with HIP
events (a) back to back in a loop by itself, (b) right after the GEMM that produces its input (x = h @ W^T, as fc1 does),
(c) after that GEMM plus ~100 MB of unrelated traffic."""
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
n = x.numel()
fb = n * (2 * 2 + 3 / 8)            # forward: read x, write y, write state
cb = n * 4                          # copy: read + write


def fwd():
    cabi.quantize_forward('gelu', x, bo, out=x, state=state)


def fwd_out_of_place():
    cabi.quantize_forward('gelu', x, bo, out=y, state=state)


def copy():
    y.copy_(x)


def negate_in_place():
    x.neg_()


def measure(kernel, before, reps=60):
    ts = []
    for i in range(reps + 10):
        before()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); kernel(); b.record()
        if i >= 10:
            ts.append((a, b))
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ts) * 1e3


def gemm():
    torch.matmul(h, w.t(), out=x)


def gemm_and_traffic():
    torch.matmul(h, w.t(), out=x)
    other2.copy_(other)


def nothing():
    pass


gemm(); torch.cuda.synchronize()
for name, kernel, nbytes in (('fewbit gelu forward, in place', fwd, fb), ('fewbit gelu forward, y <- x', fwd_out_of_place, fb), ('copy y <- x', copy, cb), ('x.neg_() (in place)', negate_in_place, cb)):
    for label, before in (('back to back', nothing), ('after the producing GEMM', gemm), ('after GEMM + 100 MB of other traffic', gemm_and_traffic)):
        us = measure(kernel, before)
        print(f'{name:32s} {label:40s} {us:7.1f} us  {nbytes / us / 1e3:7.1f} GB/s  {nbytes / us / 1e3 / 8000:.3f} of 8 TB/s', flush=True)
