import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "tests"))
import test_gpu_fused as T
from splat_slam_amd.fused import FusedMappingLoop
syn, params, cams = T._scene(n=2000, views=4)
out = []
for fuse in (True, False):
    loop = T._loop(FusedMappingLoop, syn, params, T._fresh_cams(syn, cams), range(3))
    loop.fuse_tail = fuse
    loop._ensure_state()
    snaps = []
    for it in range(3):
        loop._step([loop.viewpoints[0], loop.viewpoints[1]], iso_weight=10.0, adam=True, exposure="window")
        torch.cuda.synchronize()
        gm = loop.gaussians
        snaps.append([gm._xyz.detach().clone(), gm._features_dc.detach().clone(), gm._opacity.detach().clone(),
                      gm._scaling.detach().clone(), gm._rotation.detach().clone()])
    out.append(snaps)
names = ["xyz", "f_dc", "opacity", "scaling", "rotation"]
for it in range(3):
    for n, a, b in zip(names, out[0][it], out[1][it]):
        d = (a.double() - b.double()).abs()
        print(it, n, "max", float(d.max()), "n_diff", int((d > 0).sum()), "of", d.numel())
