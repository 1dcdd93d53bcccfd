import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from helpers import random_scene
from gpu_utils import *
for name,n,W,H,kw in [("dense",1000,64,48,dict(scale_range=(0.01,0.12))),("big",5000,128,96,dict(fx=110.,fy=104.,cx=61.4,cy=49.8,scale_range=(0.01,0.1)))]:
    inp,s=random_scene(n,seed=11,W=W,H=H,**kw); inp,s=to_fp32_inputs(inp,s)
    g=torch.Generator().manual_seed(5); wc=torch.randn(3,H,W,generator=g,dtype=torch.float64); wd=torch.randn(1,H,W,generator=g,dtype=torch.float64)
    ho,hg=run_hip(inp,s,wc,wd); ro,rg=run_oracle(inp,s,wc,wd)
    r32o,r32g=run_oracle(inp,s,wc,wd,dtype=torch.float32)
    print(name,'radii eq',torch.equal(ho[1],ro[1]),'nt diff',(ho[4]-ro[4]).abs().sum().item())
    for i,w in ((0,'color'),(2,'depth'),(3,'opac')):
        print('  ',w,'hip',outlier_report(ho[i],ro[i],1e-4),'oracle32',outlier_report(r32o[i],ro[i],1e-4))
    for k in GRAD_KEYS:
        print('  grad',k,'hip %.2e'%rel_linf(hg[k].reshape(-1),rg[k].reshape(-1)),'oracle32 %.2e'%rel_linf(r32g[k].reshape(-1),rg[k].reshape(-1)))
