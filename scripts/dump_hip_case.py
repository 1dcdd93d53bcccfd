"""GPU box: renders one config-sized view through the drop-in autograd API and stores inputs, settings, outputs and
gradients as .npz so that the comparison with the oracle can be analysed off the GPU box.
    python scripts/dump_hip_case.py <n> <camera> <scale_add> <out.npz>"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_utils import run_hip                     # noqa: E402
from oracle import raster_oracle as O             # noqa: E402
from splat_slam_amd import synthetic as syn       # noqa: E402

n, camera, scale_add, out = int(sys.argv[1]), sys.argv[2], float(sys.argv[3]), sys.argv[4]
dev = "cuda:0"
intr = syn.INTRINSICS[camera]
params = syn.room_parameters(n, seed=43, device=dev)
params["scaling"] = params["scaling"] + scale_add
cams = syn.make_views(params, 1, intr, dev, seed=43)
gm = syn.model_from_parameters(params, device=dev)
cam = cams[0]
with torch.no_grad():
    d = lambda t: t.detach().float().cpu().double()
    inp = dict(means3D=d(gm.get_xyz), means2D=torch.zeros(n, 3, dtype=torch.float64), opacities=d(gm.get_opacity),
               shs=d(gm.get_features), scales=d(gm.get_scaling), rotations=d(gm.get_rotation),
               theta=torch.zeros(3, dtype=torch.float64), rho=torch.zeros(3, dtype=torch.float64))
c = lambda t: t.detach().float().cpu().double()
f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
s = O.OracleSettings(intr["H"], intr["W"], f32(math.tan(cam.FoVx * 0.5)), f32(math.tan(cam.FoVy * 0.5)), torch.zeros(3, dtype=torch.float64),
                     1.0, c(cam.world_view_transform), c(cam.full_proj_transform), c(cam.projection_matrix), 0, c(cam.camera_center),
                     False, False)
g = torch.Generator().manual_seed(5)
wc = torch.randn(3, intr["H"], intr["W"], generator=g, dtype=torch.float64)
wd = torch.randn(1, intr["H"], intr["W"], generator=g, dtype=torch.float64)
outs, grads, keys = run_hip(inp, s, wc, wd, want_depth_keys=True)
np.savez_compressed(out, **{"in_" + k: v.numpy() for k, v in inp.items()},
                    view=s.viewmatrix.numpy(), proj=s.projmatrix.numpy(), praw=s.projmatrix_raw.numpy(), campos=s.campos.numpy(),
                    tan=np.array([s.tanfovx, s.tanfovy]), hw=np.array([intr["H"], intr["W"]]), wc=wc.numpy(), wd=wd.numpy(),
                    color=outs[0].numpy(), radii=outs[1].numpy(), depth=outs[2].numpy(), opac=outs[3].numpy(), nt=outs[4].numpy(),
                    keys=keys.numpy(), **{"g_" + k: v.numpy() for k, v in grads.items()})
print("saved", out)
