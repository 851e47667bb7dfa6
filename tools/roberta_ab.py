"""RoBERTa-base, every encoder Linear randomized (ratio 0.2): ONE process, the arms interleaved round by round on the same
box -- vanilla model | Gaussian sketch by the library's policy | with S written to memory once (tune_materialise 1) | generated
inside the product kernel (0) | the same with fp32 partial sums (round 4's data path) | Rademacher | the sampled DCT.  Step time per arm.
   python tools/roberta_ab.py fp32|bf16 [rounds]"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import fewbit
from fewbit_amd import cabi
import roberta_bench as rb

dtype = {'fp32': torch.float32, 'bf16': torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else 'fp32']
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
ids = torch.randint(5, 50000, (128, 128), generator=g).to(dev)
labels = torch.randint(0, 2, (128,), generator=g).to(dev)
vanilla = rb.build(dtype, dev)
rnd = rb.build(dtype, dev)
rb.swap_linear(rnd, 0.2, None, 'gaussian')
layers = [m for m in rnd.modules() if isinstance(m, fewbit.RandomizedLinear)]


def steps(model, n=6, warm=2):
    opt = torch.optim.SGD(model.parameters(), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        model(input_ids=ids, labels=labels).loss.backward()
        opt.step()

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def arm(kind, mem, p16):
    for m in layers:
        m.matmul = kind
    cabi.tune_sketch_materialise(mem)
    cabi.tune_sketch_partials(p16)
    return steps(rnd)


arms = {'vanilla': lambda: steps(vanilla),
        'gaussian (the policy: S from memory for 16-bit input and for fp32 input from 2048 features on)': lambda: arm('gaussian', -1, -1),
        'gaussian, S from memory (forced)': lambda: arm('gaussian', 1, -1),
        'gaussian, fused (forced)': lambda: arm('gaussian', 0, -1),
        'gaussian, fused, fp32 partial sums (round 4\'s data path)': lambda: arm('gaussian', 0, 0),
        'rademacher': lambda: arm('rademacher', -1, -1),
        "dct (the reference's sampled transform on the kernel pair fewbit_hip_sampled_dct)": lambda: arm('dct', -1, -1),
        'rademacher, fp32 partial sums (round 4\'s data path)': lambda: arm('rademacher', -1, 0)}
res = {k: [] for k in arms}
for r in range(rounds):
    for k, f in arms.items():
        res[k].append(f())
base = statistics.median(res['vanilla'])
print(f'# RoBERTa-base b128 x s128 {sys.argv[1] if len(sys.argv) > 1 else "fp32"}, ms per step: median of {rounds} interleaved rounds of 6 steps (min..max), x vanilla')
for k, v in res.items():
    print(f'{k:78s} {statistics.median(v):7.2f} ({min(v):7.2f}..{max(v):7.2f})  {statistics.median(v) / base:.3f}x', flush=True)
