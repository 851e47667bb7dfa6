"""driver for rocprofv3: N x (fwd, bwd) of gelu k=3 4096x4096 bf16 through the C-ABI"""
import sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from fewbit_amd import cabi
from tests.helpers import from_raw
z = np.load('tests/golden/quantize_ref.npz')
dev = 'cuda'
n = 4096 * 4096
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
b = from_raw(z['gelu03_bf16_borders'], torch.bfloat16).to(dev); l = from_raw(z['gelu03_bf16_levels'], torch.bfloat16).to(dev)
x = torch.randn(n, device=dev).to(torch.bfloat16); gy = torch.randn(n, device=dev).to(torch.bfloat16)
y = torch.empty_like(x); gx = torch.empty_like(x)
st = torch.empty(cabi.state_nbytes(n, 3), dtype=torch.uint8, device=dev)
for _ in range(iters):
    cabi.quantize_forward('gelu', x, b, out=y, state=st)
    cabi.quantize_backward(gy, st, l, out=gx)
torch.cuda.synchronize()
print('done')
