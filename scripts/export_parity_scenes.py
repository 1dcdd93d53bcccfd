#!/usr/bin/env python3
"""CUDA-box parity kit, part 1 (runs anywhere; uses the GPU only if one is present).

The rasterizer this repository replaces -- rmurai0610/diff-gaussian-rasterization-w-pose @ 43e21bf (+ near-plane patch,
/root/reference/README.md:88-92) -- is absent from /root/reference and is CUDA, so the oracle is pinned by known answers
and finite differences only.  This script writes the parity scenes of tests/test_gpu_parity.py (+ one
config-sized view) as .npz files that hold, for each scene,

  args_fwd_*   the arguments of upstream's `_C.rasterize_gaussians` in ITS order:
               bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
               projmatrix, projmatrix_raw, tanfovx, tanfovy, image_height, image_width, sh, degree, campos, prefiltered, debug
  args_bwd_*   the two upstream gradients `_C.rasterize_gaussians_backward` consumes: dL_dcolor, dL_ddepth
  oracle_*     outputs and gradients of oracle/raster_oracle.py (fp64), named like the 5-tuple / the autograd inputs
  oracle_upj_* the pose gradients with UPSTREAM_POSE_JACOBIAN switched on (the one documented open choice)
  hip_*        the same from this repository's HIP build (only when run on an MI355X)

and scripts/compare_with_upstream_cuda.py (part 2, ~60 lines, needs nothing from this repository) replays them through the
real CUDA extension on an NVIDIA machine and prints the relative errors.

    python scripts/export_parity_scenes.py [out_dir] [--big] [--only=tiny,two,...]
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import random_scene                       # noqa: E402
from gpu_utils import GRAD_KEYS, run_oracle, to_fp32_inputs   # noqa: E402
from oracle import raster_oracle as O                   # noqa: E402


def export(name, inp, s, out_dir, seed=5):
    H, W = int(s.image_height), int(s.image_width)
    g = torch.Generator().manual_seed(seed)
    wc = torch.randn(3, H, W, generator=g, dtype=torch.float64)
    wd = torch.randn(1, H, W, generator=g, dtype=torch.float64)
    f32 = lambda t: t.detach().float().cpu().numpy()
    n = inp["means3D"].shape[0]
    d = {"args_fwd_bg": f32(s.bg), "args_fwd_means3D": f32(inp["means3D"]), "args_fwd_colors_precomp": np.zeros((0,), np.float32),
         "args_fwd_opacities": f32(inp["opacities"]), "args_fwd_scales": f32(inp["scales"]), "args_fwd_rotations": f32(inp["rotations"]),
         "args_fwd_scale_modifier": np.float32(s.scale_modifier), "args_fwd_cov3D_precomp": np.zeros((0,), np.float32),
         "args_fwd_viewmatrix": f32(s.viewmatrix), "args_fwd_projmatrix": f32(s.projmatrix),
         "args_fwd_projmatrix_raw": f32(s.projmatrix_raw), "args_fwd_tanfovx": np.float32(s.tanfovx),
         "args_fwd_tanfovy": np.float32(s.tanfovy), "args_fwd_image_height": np.int32(H), "args_fwd_image_width": np.int32(W),
         "args_fwd_sh": f32(inp["shs"]), "args_fwd_degree": np.int32(s.sh_degree), "args_fwd_campos": f32(s.campos),
         "args_fwd_prefiltered": np.bool_(False), "args_fwd_debug": np.bool_(False),
         "args_bwd_dL_dcolor": f32(wc), "args_bwd_dL_ddepth": f32(wd)}
    names = ["color", "radii", "depth", "opacity", "n_touched"]
    out, grads = run_oracle(inp, s, wc, wd, dtype=torch.float64)
    for k, v in zip(names, out):
        d["oracle_" + k] = v.numpy()
    for k in GRAD_KEYS:
        d["oracle_grad_" + k] = grads[k].numpy()
    O.UPSTREAM_POSE_JACOBIAN = True
    try:
        _, g_up = run_oracle(inp, s, wc, wd, dtype=torch.float64)
    finally:
        O.UPSTREAM_POSE_JACOBIAN = False
    d["oracle_upj_grad_theta"], d["oracle_upj_grad_rho"] = g_up["theta"].numpy(), g_up["rho"].numpy()
    if torch.cuda.is_available():
        from gpu_utils import run_hip
        hout, hg = run_hip(inp, s, wc, wd)
        for k, v in zip(names, hout):
            d["hip_" + k] = v.numpy()
        for k in GRAD_KEYS:
            d["hip_grad_" + k] = hg[k].numpy()
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, "N =", n, "visible =", int((out[1] > 0).sum()))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_dir = args[0] if args else os.path.join(ROOT, "gpurun_out", "parity_kit")
    os.makedirs(out_dir, exist_ok=True)
    from test_gpu_parity import CASES
    only = next((a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--only=")), None)
    for name, n, W, H, kw in CASES:
        if kw.get("sh_degree", 0) > 0 or n > 1500 or (only is not None and name not in only):
            continue                                     # degree-0 scenes small enough to mail around
        inp, s = random_scene(n, seed=11, W=W, H=H, **kw)
        inp, s = to_fp32_inputs(inp, s)
        export(name, inp, s, out_dir)
    if "--big" in sys.argv:                              # one config-sized view (640x480, 300 k Gaussians): needs the GPU for knn
        from splat_slam_amd import synthetic as syn
        intr = syn.INTRINSICS["metric"]
        dev = "cuda:0" if torch.cuda.is_available() else "cpu"
        knn = None if dev != "cpu" else (lambda p: torch.full((p.shape[0],), 0.018 ** 2))
        params = syn.room_parameters(300000, seed=43, device=dev, knn_fn=knn)
        w2c = syn.orbit_w2c(0, 16).double()
        s = O.make_settings(w2c, intr["fx"], intr["fy"], intr["cx"], intr["cy"], intr["W"], intr["H"], dtype=torch.float64)
        c = lambda t: t.detach().float().cpu().double()
        inp = dict(means3D=c(params["xyz"]), means2D=torch.zeros(300000, 3, dtype=torch.float64),
                   opacities=c(torch.sigmoid(params["opacity"])), shs=c(params["f_dc"]), scales=c(torch.exp(params["scaling"])),
                   rotations=c(torch.nn.functional.normalize(params["rotation"])), theta=torch.zeros(3, dtype=torch.float64),
                   rho=torch.zeros(3, dtype=torch.float64))
        inp, s = to_fp32_inputs(inp, s)
        export("room_300k_640x480", inp, s, out_dir)


if __name__ == "__main__":
    main()
