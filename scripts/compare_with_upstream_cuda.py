#!/usr/bin/env python3
"""CUDA-box parity kit, part 2.  Needs ONLY: an NVIDIA GPU, torch, numpy and the pinned upstream extension
(rmurai0610/diff-gaussian-rasterization-w-pose @ 43e21bf with the near-plane patch of Splat-SLAM's README.md:88-92
installed as `diff_gaussian_rasterization`).  Replays every scene written by scripts/export_parity_scenes.py through it:

    python compare_with_upstream_cuda.py parity_kit/*.npz

and prints, per output / gradient, max|cuda - x| / max|x| for x = the oracle (and the HIP build when its results are in
the file).  1e-4 is the bar (BASELINE.json north_star).  If only grad_theta / grad_rho miss it, look at the
`oracle_upj_*` column: it is the same gradient with UPSTREAM_POSE_JACOBIAN on."""
import sys

import numpy as np
import torch
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

rel = lambda a, b: float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(np.abs(b).max(), 1e-30))
for path in sys.argv[1:]:
    Z = np.load(path)
    t = lambda k, grad=False: torch.from_numpy(Z["args_fwd_" + k]).cuda().requires_grad_(grad)
    rs = GaussianRasterizationSettings(
        image_height=int(Z["args_fwd_image_height"]), image_width=int(Z["args_fwd_image_width"]), tanfovx=float(Z["args_fwd_tanfovx"]),
        tanfovy=float(Z["args_fwd_tanfovy"]), bg=t("bg"), scale_modifier=float(Z["args_fwd_scale_modifier"]), viewmatrix=t("viewmatrix"),
        projmatrix=t("projmatrix"), projmatrix_raw=t("projmatrix_raw"), sh_degree=int(Z["args_fwd_degree"]), campos=t("campos"),
        prefiltered=False, debug=False)
    x = {k: t(k, True) for k in ("means3D", "opacities", "sh", "scales", "rotations")}
    means2D = torch.zeros_like(x["means3D"], requires_grad=True)
    theta = torch.zeros(3, device="cuda", requires_grad=True)
    rho = torch.zeros(3, device="cuda", requires_grad=True)
    color, radii, depth, opacity, n_touched = GaussianRasterizer(raster_settings=rs)(
        means3D=x["means3D"], means2D=means2D, shs=x["sh"], colors_precomp=None, opacities=x["opacities"], scales=x["scales"],
        rotations=x["rotations"], cov3D_precomp=None, theta=theta, rho=rho)
    wc, wd = torch.from_numpy(Z["args_bwd_dL_dcolor"]).cuda(), torch.from_numpy(Z["args_bwd_dL_ddepth"]).cuda()
    ((color * wc).sum() + (depth * wd).sum()).backward()
    got = {"color": color, "radii": radii, "depth": depth, "opacity": opacity, "n_touched": n_touched,
           "grad_means3D": x["means3D"].grad, "grad_means2D": means2D.grad, "grad_opacities": x["opacities"].grad,
           "grad_shs": x["sh"].grad, "grad_scales": x["scales"].grad, "grad_rotations": x["rotations"].grad,
           "grad_theta": theta.grad, "grad_rho": rho.grad}
    print(f"== {path}")
    for k, v in got.items():
        v = v.detach().cpu().numpy()
        row = [f"{k:16s}"]
        for who in ("oracle_", "oracle_upj_", "hip_"):
            row.append(f"{who}{rel(v.reshape(-1), Z[who + k].reshape(-1)):.2e}" if (who + k) in Z.files else " " * 12)
        print("  ".join(row))
