"""fp32 RoBERTa-base, Gaussian ratio 0.2: which layers may read S from memory -- none, all, the 3072-wide ones only (the policy), the
768-wide ones only; arms interleaved in one process"""
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


def build_model():
    if not os.environ.get('LARGE'):
        return rb.build(torch.float32, dev)
    from transformers import RobertaConfig, RobertaForSequenceClassification      # RoBERTa-large: 1024 / 4096 wide, 24 layers
    torch.manual_seed(0)
    cfg = RobertaConfig(num_labels=2, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    return RobertaForSequenceClassification(cfg).to(device=dev, dtype=torch.float32).train()


def make(kind):
    m = build_model()
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


models = {k: make(k) for k in (None, 'gaussian')}
arms = [('vanilla', None, -1), ('fused everywhere', 'gaussian', 0), ('S from memory everywhere', 'gaussian', 1),
        ('S from memory, 3072 wide only (policy)' if not os.environ.get('LARGE') else 'S from memory, 4096 wide only (policy)', 'gaussian', -1),
        ('S from memory, 768 wide only' if not os.environ.get('LARGE') else 'S from memory, 1024 wide only', 'gaussian', 2)]
if os.environ.get('ARMS'):                       # e.g. ARMS=0,2: vanilla + the arms with those tune values
    keep = {int(x) for x in os.environ['ARMS'].split(',')}
    arms = [a for a in arms if a[1] is None or a[2] in keep]
res = {a[0]: [] for a in arms}
for r in range(rounds):
    for name, kind, mem in arms:
        cabi.tune_sketch_materialise(mem)
        res[name].append(steps(*models[kind]))
cabi.tune_sketch_materialise(-1)
v = statistics.median(res['vanilla'])
for name, _, _ in arms:
    m = statistics.median(res[name])
    print(f'{name:40s} {m:8.2f} ms per step  {m / v:.3f}x vanilla   rounds: ' + ' '.join(f'{x:.2f}' for x in res[name]))
