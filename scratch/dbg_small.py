import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from fewbit_amd import cabi
from fewbit_amd.store import store
dev='cuda'
b,_=store.get('gelu',3,dev,torch.float32); inner=b[1:-1].contiguous()
xs=torch.tensor([1e-45,1e-40,1e-38,1e-30,1e-20,1e-10,1e-8,1e-7,1e-6,1e-5,1e-4,1e-3,1e-2,0.1,0.5,1.0,2.0,5.0]*64, device=dev)
y,_=cabi.quantize_forward('gelu', xs, inner)
xd=xs.double(); ex=(xd*0.5*(1+torch.erf(xd*0.7071067811865476))).float()
at=torch.nn.functional.gelu(xs)
for i in range(18):
    print(f'x={xs[i].item():.3e} y={y[i].item():.9e} exact={ex[i].item():.9e} aten={at[i].item():.9e} ulp={(y.view(torch.int32)[i]-ex.view(torch.int32)[i]).item()}')
