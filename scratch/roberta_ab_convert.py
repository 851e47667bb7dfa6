"""fp32 RoBERTa-base, randomized linear layers (ratio 0.2): the library's policy (fp32 input rounded to bf16 by its own pass when p > 1280)
against staging fp32 directly in the product kernel (tune_sketch_convert(0)); arms interleaved in one process"""
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
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def make(kind):
    m = rb.build(torch.float32, dev)
    if kind:
        rb.swap_linear(m, 0.2, None, kind)
    return m, torch.optim.SGD(m.parameters(), lr=1e-4)


def steps(m, opt, n=6, warm=2):
    def step():
        opt.zero_grad(set_to_none=True)
        m(input_ids=ids, labels=labels).loss.backward()
        opt.step()
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


models = {k: make(k) for k in (None, 'gaussian', 'rademacher')}
arms = [('vanilla', None, -1), ('gaussian policy', 'gaussian', -1), ('gaussian fp32 staged directly', 'gaussian', 0),
        ('rademacher policy', 'rademacher', -1), ('rademacher fp32 staged directly', 'rademacher', 0)]
res = {a[0]: [] for a in arms}
for r in range(rounds):
    for name, kind, conv in arms:
        cabi.tune_sketch_convert(conv)
        res[name].append(steps(*models[kind]))
cabi.tune_sketch_convert(-1)
v = statistics.median(res['vanilla'])
for name, _, _ in arms:
    m = statistics.median(res[name])
    print(f'{name:34s} {m:8.2f} ms per step  {m / v:.3f}x vanilla   rounds: ' + ' '.join(f'{x:.2f}' for x in res[name]))
