"""Why did the bf16 DCT arm of tools/convergence_demo.py reach NaN at lr = 1.0?  The same student with the DCT layers on the kernel pair and on
the torch.fft formulation, several seeds: first step whose loss or any gradient is non-finite, and the largest |grad| seen before it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import fewbit
from fewbit_amd import linear
dev, dtype = 'cuda:0', torch.bfloat16
ROWS, DIN, HID, DOUT = 4096, 256, 1024, 64
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
torch.manual_seed(1)
teacher = nn.Sequential(nn.Linear(DIN, HID), nn.Tanh(), nn.Linear(HID, DOUT)).to(dev).float()
for native in (True, False):
    for seed in (0, 1, 2):
        prev = linear.use_native_sketch(native)
        torch.manual_seed(seed)
        model = nn.Sequential(nn.Linear(DIN, HID), fewbit.GELU(bits=3), fewbit.RandomizedLinear(HID, HID, proj_dim_ratio=0.2, matmul='dct'), fewbit.GELU(bits=3),
                              fewbit.RandomizedLinear(HID, HID, proj_dim_ratio=0.2, matmul='dct'), fewbit.GELU(bits=3), nn.Linear(HID, DOUT)).to(dev).to(dtype)
        opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9)
        g = torch.Generator(device=dev).manual_seed(7)
        bad, gmax, last = None, 0.0, None
        for step in range(300):
            x = torch.randn(ROWS, DIN, device=dev, generator=g)
            with torch.no_grad():
                y = teacher(x)
            loss = ((model(x.to(dtype)).float() - y) ** 2).mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            gm = max(float(p.grad.float().abs().max()) for p in model.parameters())
            if not (torch.isfinite(loss) and gm == gm and gm != float('inf')):
                bad = step
                break
            gmax, last = max(gmax, gm), float(loss)
            opt.step()
        print(f"{'kernel pair' if native else 'torch.fft  '} seed {seed}: first non-finite step {bad}, last finite loss {last:.5f}, largest |grad| before {gmax:.3e}")
        linear.use_native_sketch(prev)
