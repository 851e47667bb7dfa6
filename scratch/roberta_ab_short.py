"""two arms of scratch/roberta_ab.py only (fp32 RoBERTa-base, Gaussian: fused against S from memory), 2 rounds: classifies a box in a minute"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import fewbit
from fewbit_amd import cabi
import roberta_bench as rb
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
ids = torch.randint(5, 50000, (128, 128), generator=g).to(dev)
labels = torch.randint(0, 2, (128,), generator=g).to(dev)
rnd = rb.build(torch.float32, dev)
rb.swap_linear(rnd, 0.2, None, 'gaussian')
opt = torch.optim.SGD(rnd.parameters(), lr=1e-4)


def steps(n=6, warm=2):
    def step():
        opt.zero_grad(set_to_none=True)
        rnd(input_ids=ids, labels=labels).loss.backward()
        opt.step()
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {0: [], 1: []}
for r in range(2):
    for mem in (0, 1):
        cabi.tune_sketch_materialise(mem)
        res[mem].append(steps())
f, m = statistics.median(res[0]), statistics.median(res[1])
print(f'fp32 RoBERTa-base, Gaussian ratio 0.2: fused {f:.2f} ms, S from memory {m:.2f} ms per step -> S from memory is {"AHEAD" if m < f else "BEHIND"} by {abs(m - f) / f * 100:.1f} %')
