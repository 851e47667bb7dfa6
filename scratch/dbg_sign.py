import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fewbit, oracle
from fewbit_amd import cabi
g = torch.Generator().manual_seed(3)
for dtype in (torch.float32, torch.bfloat16, torch.float16):
    x = (torch.randn(4099, generator=g) * 2).to(dtype)
    gy = torch.randn(4099, generator=g).to(dtype)
    _, bits_o = oracle.stepwise1_forward('relu', x)
    gr_o = oracle.stepwise1_backward('relu', gy, bits_o)
    it = torch.int16 if dtype != torch.float32 else torch.int32
    yd, st = cabi.stepwise1_forward('relu', x.cuda())
    print(dtype, 'state equal', torch.equal(st.cpu(), bits_o))
    gx = cabi.stepwise1_backward('relu', gy.cuda(), st)
    d = (gx.cpu().view(it) != gr_o.view(it)).nonzero().flatten()
    print('  cabi backward mismatches', d.tolist(), [(x[i].item(), gy[i].item(), gx[i].item(), gr_o[i].item()) for i in d[:4]])
    for label, mk in (('mul', lambda t: t * 1.0), ('clone', lambda t: t.clone())):
        xr = x.cuda().requires_grad_()
        yr = torch.ops.fewbit.relu(mk(xr))
        yr.backward(gy.cuda())
        d = (xr.grad.cpu().view(it) != gr_o.view(it)).nonzero().flatten()
        print('  autograd via', label, 'mismatches', d.tolist(), [(x[i].item(), gy[i].item(), xr.grad[i].item(), gr_o[i].item()) for i in d[:4]])
    m = (gy.cuda() * 1.0)
    print('  gy*1.0 bit-equal gy:', torch.equal(m.cpu().view(it), gy.view(it)))
    z = torch.zeros(4099, dtype=dtype, device='cuda') * gy.cuda()
    zz = (z * 1.0)
    print('  (0*gy)*1.0 sign kept:', torch.equal(zz.cpu().view(it), z.cpu().view(it)), (zz.cpu().view(it) != z.cpu().view(it)).nonzero().flatten().tolist()[:5])
