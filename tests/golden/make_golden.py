#!/usr/bin/env python3
"""Generates tests/golden/reference_vectors.npz + reference_config.json by IMPORTING the Python pieces of the
reference that run on CPU.  Run in the build container only (/root/reference is not on the GPU
box); the outputs are data (inputs + expected outputs), committed next to this script.

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.modules.setdefault("cv2", types.ModuleType("cv2"))          # loss_utils / slam_utils import it, never call it here
torch.Tensor.cuda = lambda self, *a, **k: self                  # get_loss_mapping_rgbd calls .cuda() on the gt image

from thirdparty.gaussian_splatting.utils.graphics_utils import focal2fov, getProjectionMatrix2, getWorld2View2  # noqa: E402
from thirdparty.gaussian_splatting.utils.general_utils import helper, inverse_sigmoid  # noqa: E402
from thirdparty.gaussian_splatting.utils.general_utils import rotation_matrix_to_quaternion, quaternion_multiply  # noqa: E402
from thirdparty.gaussian_splatting.utils.sh_utils import RGB2SH, eval_sh  # noqa: E402
from thirdparty.gaussian_splatting.utils.image_utils import psnr  # noqa: E402
from thirdparty.monogs.utils.pose_utils import SE3_exp, update_pose  # noqa: E402
from thirdparty.monogs.utils.camera_utils import Camera  # noqa: E402
from thirdparty.monogs.utils.slam_utils import get_loss_mapping, get_median_depth  # noqa: E402
from thirdparty.glorie_slam import config as ref_config  # noqa: E402

g = torch.Generator().manual_seed(43)
out = {}

# ---- G1: projection / view / full-projection / camera-centre for the 4 dataset intrinsics x 8 poses
INTR = {   # W, H, fx, fy, cx, cy  (after datasets.py:94-107 scaling)
    "default640x480": (640, 480, 400.0, 400.0, 319.5, 239.5),
    "replica640x320": (640, 320, 320.0, 600.0 * 320.0 / 680.0, 599.5 * 640.0 / 1200.0, 339.5 * 320.0 / 680.0),
    "tum512x384": (512, 384, 517.3 * 512 / 640.0, 516.5 * 384 / 480.0, 318.6 * 512 / 640.0, 255.3 * 384 / 480.0),
    "scannet320x240": (320, 240, 577.59 * 320 / 640.0, 578.73 * 240 / 480.0, 318.9 * 320 / 640.0, 242.68 * 240 / 480.0),
}
poses = []
for _ in range(8):
    w = 0.4 * torch.randn(3, generator=g)
    R = torch.linalg.matrix_exp(torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]))
    t = torch.randn(3, generator=g)
    poses.append((R, t))
out["g1_pose_R"] = torch.stack([p[0] for p in poses]).numpy()
out["g1_pose_T"] = torch.stack([p[1] for p in poses]).numpy()
for name, (W, H, fx, fy, cx, cy) in INTR.items():
    P = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
    out[f"g1_{name}_intr"] = np.array([W, H, fx, fy, cx, cy], dtype=np.float64)
    out[f"g1_{name}_proj"] = P.numpy()
    out[f"g1_{name}_fov"] = np.array([focal2fov(fx, W), focal2fov(fy, H)])
    views, fulls, centers = [], [], []
    for R, t in poses:
        cam = Camera(0, None, None, torch.eye(4), P, fx, fy, cx, cy, focal2fov(fx, W), focal2fov(fy, H), H, W, device="cpu")
        cam.update_RT(R, t)
        views.append(cam.world_view_transform.numpy())
        fulls.append(cam.full_proj_transform.numpy())
        centers.append(cam.camera_center.numpy())
    out[f"g1_{name}_view"] = np.stack(views)
    out[f"g1_{name}_full"] = np.stack(fulls)
    out[f"g1_{name}_center"] = np.stack(centers)

# ---- G2: SE3_exp for 32 taus incl. the small-angle branch
taus = torch.randn(32, 6, generator=g) * 0.5
taus[28:, 3:] *= 1e-6
taus[31] = 0
out["g2_tau"] = taus.numpy()
out["g2_T"] = torch.stack([SE3_exp(t) for t in taus]).numpy()

# ---- G3: update_pose before/after
P = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=400.0, fy=400.0, cx=319.5, cy=239.5, W=640, H=480).transpose(0, 1)
before, after, conv = [], [], []
for i in range(6):
    cam = Camera(1, None, None, torch.eye(4), P, 400.0, 400.0, 319.5, 239.5, 1.0, 1.0, 480, 640, device="cpu")
    R, t = poses[i]
    cam.update_RT(R, t)
    with torch.no_grad():
        cam.cam_trans_delta.copy_(taus[i, :3] * (1e-5 if i == 5 else 0.1))
        cam.cam_rot_delta.copy_(taus[i, 3:] * (1e-5 if i == 5 else 0.1))
    before.append(torch.cat([cam.cam_trans_delta.detach(), cam.cam_rot_delta.detach()]).numpy().copy())
    c = update_pose(cam)
    conv.append(bool(c))
    M = np.eye(4, dtype=np.float32)
    M[:3, :3] = cam.R.detach().numpy()
    M[:3, 3] = cam.T.detach().numpy()
    after.append(M)
out["g3_tau"] = np.stack(before)
out["g3_w2c_after"] = np.stack(after)
out["g3_converged"] = np.array(conv)

# ---- G4: learning-rate schedule
steps = [0, 1, 100, 1050, 30000, 40000]
out["g4_steps"] = np.array(steps)
out["g4_lr"] = np.array([helper(s, lr_init=0.00016 * 6.0, lr_final=0.0000016 * 6.0, lr_delay_mult=0.01, max_steps=30000) for s in steps])

# ---- G5: mapping loss value + autograd gradients
class _VP:
    pass


cfg = {"Training": {"alpha": 0.8, "rgb_boundary_threshold": 0.01, "ssim_loss": False}, "opt_params": {"lambda_dssim": 0.2}}
for tag, init in (("g5a", False), ("g5b", True)):
    H, W = 32, 48
    image = torch.rand(3, H, W, generator=g).requires_grad_(True)
    depth = (3 * torch.rand(1, H, W, generator=g)).requires_grad_(True)
    gt = torch.rand(3, H, W, generator=g)
    gt[:, :4, :7] = 0.0                         # black border pixels: masked out
    gtd = 3 * torch.rand(H, W, generator=g)
    gtd[10:14, 20:30] = 0.0                     # invalid depth
    vp = _VP()
    vp.original_image = gt
    vp.depth = gtd.numpy()
    vp.exposure_a = torch.tensor([0.07], requires_grad=True)
    vp.exposure_b = torch.tensor([-0.03], requires_grad=True)
    loss = get_loss_mapping(cfg, image, depth, vp, None, initialization=init)
    loss.backward()
    out[f"{tag}_image"], out[f"{tag}_depth"], out[f"{tag}_gt"], out[f"{tag}_gtd"] = image.detach().numpy(), depth.detach().numpy(), gt.numpy(), gtd.numpy()
    out[f"{tag}_loss"] = np.array(loss.item())
    out[f"{tag}_dimage"], out[f"{tag}_ddepth"] = image.grad.numpy(), depth.grad.numpy()
    out[f"{tag}_da"] = np.array(0.0 if vp.exposure_a.grad is None else vp.exposure_a.grad.item())
    out[f"{tag}_db"] = np.array(0.0 if vp.exposure_b.grad is None else vp.exposure_b.grad.item())
d = 3 * torch.rand(1, 16, 16, generator=g)
o = torch.rand(1, 16, 16, generator=g)
out["g5_med_depth"], out["g5_med_opacity"] = d.numpy(), o.numpy()
out["g5_median"] = np.array(get_median_depth(d, o * 0 + 0.96).item())
out["g5_median_masked"] = np.array(get_median_depth(d, torch.where(o > 0.5, o * 0 + 0.99, o * 0)).item())

# ---- G6: psnr
a, b = torch.rand(1, 3, 20, 20, generator=g), torch.rand(1, 3, 20, 20, generator=g)
out["g6_a"], out["g6_b"], out["g6_psnr"] = a.numpy(), b.numpy(), psnr(a, b).numpy()

# ---- G7: SH evaluation deg 0..3, RGB2SH
dirs = torch.randn(16, 3, generator=g)
dirs = dirs / dirs.norm(dim=1, keepdim=True)
sh = torch.randn(16, 3, 16, generator=g)
out["g7_dirs"], out["g7_sh"] = dirs.numpy(), sh.numpy()
for deg in range(4):
    out[f"g7_eval_deg{deg}"] = eval_sh(deg, sh, dirs).numpy()
rgb = torch.rand(5, 3, generator=g)
out["g7_rgb"], out["g7_rgb2sh"] = rgb.numpy(), RGB2SH(rgb).numpy()
x = torch.rand(9, generator=g) * 0.98 + 0.01
out["g7_x"], out["g7_inverse_sigmoid"] = x.numpy(), inverse_sigmoid(x).numpy()

# ---- G8: rotation helpers of the map-deformation step
Rs = torch.stack([p[0] for p in poses])
qs = torch.stack([rotation_matrix_to_quaternion(R[None])[0] for R in Rs])
out["g8_R"], out["g8_quat"] = Rs.numpy(), qs.numpy()
q2 = torch.randn(8, 4, generator=g)
q2 = q2 / q2.norm(dim=1, keepdim=True)
out["g8_q2"], out["g8_qmul"] = q2.numpy(), quaternion_multiply(qs, q2).numpy()

np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)

# ---- G9: flattened hyper-parameters of configs/Replica/office0.yaml (inheritance resolved by the reference's loader)
_cwd = os.getcwd()
os.chdir(REF)            # inherit_from paths in the YAMLs are relative to the reference root
cfg = ref_config.load_config("configs/Replica/office0.yaml", "configs/splat_slam.yaml")
os.chdir(_cwd)
keep = {"mapping": cfg["mapping"], "cam": cfg["cam"], "setup_seed": cfg["setup_seed"]}
json.dump(keep, open(os.path.join(HERE, "reference_config.json"), "w"), indent=1, sort_keys=True, default=str)
print("wrote", len(out), "arrays")
