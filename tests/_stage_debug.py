import sys, time, os
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else '.')
t0=time.time()
def log(*a):
    print("[%.1fs]"%(time.time()-t0), *a, flush=True)
import torch
log("torch imported")
from splat_slam_amd import synthetic as syn
from splat_slam_amd.mapper import MappingLoop, PipelineParams
from splat_slam_amd.renderer import render
dev=torch.device('cuda:0')
N=int(sys.argv[1]) if len(sys.argv)>1 else 300000
gen=torch.Generator().manual_seed(43)
pts=syn.room_points(N,gen).to(dev)
log("points")
from simple_knn._C import distCUDA2
d=distCUDA2(pts); torch.cuda.synchronize()
log("knn", d.mean().item())
params=syn.room_parameters(N,seed=43,device=dev); torch.cuda.synchronize()
log("params")
intr=syn.INTRINSICS['metric']
cams=syn.make_views(params,4,intr,dev); torch.cuda.synchronize()
log("views")
loop=MappingLoop(syn.DEFAULT_CONFIG,device=dev)
loop.gaussians=syn.model_from_parameters(params,device=dev)
loop.viewpoints={c.uid:c for c in cams}
loop.current_window=[0,1]
loop.build_keyframe_optimizers()
log("loop built")
pkg=render(cams[0],loop.gaussians,PipelineParams(),loop.background); torch.cuda.synchronize()
log("render fwd", (pkg['radii']>0).sum().item())
loop.map(loop.current_window,iters=1); torch.cuda.synchronize()
log("map iter 1")
t=time.time()
for _ in range(5): loop.map(loop.current_window,iters=1)
torch.cuda.synchronize()
log("5 map iters: %.2f ms/iter"%((time.time()-t)/5*1e3))
