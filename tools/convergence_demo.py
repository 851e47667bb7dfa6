"""A small end-to-end check that the drop-in trains: a student MLP (256 -> 1024 -> 1024 -> 64, 4096 rows per step, bf16 or fp32) is
fitted to a fixed random teacher for a few hundred optimizer steps, once with torch's own layers and once per few-bit configuration:

    vanilla                         nn.Linear + nn.GELU
    gelu 3-bit                      fewbit.GELU(bits=3)                          (exact forward, 3-bit backward)
    gelu 3-bit + linear <kind>      ... and every hidden nn.Linear -> fewbit.RandomizedLinear(proj_dim_ratio=0.2, matmul=<kind>)

Prints the loss after 0 / 100 / 200 / 300 steps and the peak memory per arm (one JSON line at the end).  Not a benchmark: evidence that
forward values, the quantized backward and the randomized weight gradients compose into a model that learns at the vanilla rate.

    python3 tools/convergence_demo.py [bf16|fp32] [steps=300] [lr]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import fewbit  # noqa: E402

dtype = {'bf16': torch.bfloat16, 'fp32': torch.float32}[sys.argv[1] if len(sys.argv) > 1 else 'fp32']
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = 'cuda:0'
ROWS, DIN, HID, DOUT = 4096, 256, 1024, 64
LR = float(sys.argv[3]) if len(sys.argv) > 3 else (1e-3 if dtype == torch.float32 else 0.7)         # Adam for fp32 weights, SGD + momentum for bf16 weights


def build(kind):
    torch.manual_seed(0)
    act = (lambda: nn.GELU()) if kind == 'vanilla' else (lambda: fewbit.GELU(bits=3))
    if kind in ('vanilla', 'gelu'):
        lin = lambda i, o: nn.Linear(i, o)                                                       # noqa: E731
    else:
        lin = lambda i, o: fewbit.RandomizedLinear(i, o, proj_dim_ratio=0.2, matmul=kind)       # noqa: E731
    model = nn.Sequential(nn.Linear(DIN, HID), act(), lin(HID, HID), act(), lin(HID, HID), act(), nn.Linear(HID, DOUT))
    return model.to(dev).to(dtype)


torch.manual_seed(1)
teacher = nn.Sequential(nn.Linear(DIN, HID), nn.Tanh(), nn.Linear(HID, DOUT)).to(dev).float()
out = {'dtype': str(dtype), 'rows_per_step': ROWS, 'steps': steps, 'arms': {}}
for name, kind in (('vanilla', 'vanilla'), ('gelu 3-bit', 'gelu'), ('gelu 3-bit + linear dct', 'dct'), ('gelu 3-bit + linear rademacher', 'rademacher'),
                   ('gelu 3-bit + linear gaussian', 'gaussian')):
    model = build(kind)
    opt = torch.optim.Adam(model.parameters(), lr=LR) if dtype == torch.float32 else torch.optim.SGD(model.parameters(), lr=LR, momentum=0.9)
    g = torch.Generator(device=dev).manual_seed(7)
    torch.cuda.reset_peak_memory_stats()
    losses = {}
    for step in range(steps + 1):
        x = torch.randn(ROWS, DIN, device=dev, generator=g)
        with torch.no_grad():
            y = teacher(x)
        loss = ((model(x.to(dtype)).float() - y) ** 2).mean()
        if step % 100 == 0:
            losses[step] = round(float(loss.detach()), 5)
        if step == steps:
            break
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    out['arms'][name] = {'loss': losses, 'peak_MiB': round(torch.cuda.max_memory_allocated() / 2**20, 1)}
    print(f'{name:<34} loss {losses}   peak {out["arms"][name]["peak_MiB"]} MiB')
    del model, opt
print(json.dumps(out))
