"""Synthetic mapping-only feed: the "room" scene, its cameras and the keyframe stream that stands
in for tracker + DepthVideo + dataset, none of which exist on the GPU box (no datasets, no network)."""
import math

import numpy as np
import torch

from splat_slam_amd.camera import Camera, focal2fov, getProjectionMatrix2
from splat_slam_amd.gaussian_model import GaussianModel, OptParams, RGB2SH

# hot-path part of /root/reference/configs/splat_slam.yaml:22-86
DEFAULT_CONFIG = {
    "mapping": {
        "BA": False, "pcd_downsample": 32, "pcd_downsample_init": 16, "adaptive_pointsize": True, "point_size": 0.05,
        "final_refine_iters": 26000, "move_points": True,
        "Training": {"ssim_loss": False, "gt_camera": False, "alpha": 0.80, "init_itr_num": 1050,
                     "init_gaussian_update": 100, "init_gaussian_reset": 500, "init_gaussian_th": 0.005,
                     "init_gaussian_extent": 30, "mapping_itr_num": 60, "gaussian_update_every": 150,
                     "gaussian_update_offset": 50, "gaussian_th": 0.7, "gaussian_extent": 1.0, "gaussian_reset": 2001,
                     "size_threshold": 20, "window_size": 10, "pose_window": 5, "edge_threshold": 4,
                     "rgb_boundary_threshold": 0.01, "kf_translation": 0.04, "kf_min_translation": 0.02,
                     "kf_overlap": 0.95, "prune_mode": "slam", "spherical_harmonics": False,
                     "lr": {"cam_rot_delta": 0.003, "cam_trans_delta": 0.001}},
        "opt_params": {"position_lr_init": 0.00016, "position_lr_final": 0.0000016, "position_lr_delay_mult": 0.01,
                       "position_lr_max_steps": 30000, "feature_lr": 0.0025, "opacity_lr": 0.05, "scaling_lr": 0.001,
                       "rotation_lr": 0.001, "percent_dense": 0.01, "lambda_dssim": 0.2, "densify_from_iter": 500,
                       "densify_grad_threshold": 0.0002},
    }
}

# (i) the metric's 640x480 (configs/splat_slam.yaml:128-129); (ii) Replica after datasets.py:94-104 scaling
INTRINSICS = {
    "metric": dict(W=640, H=480, fx=400.0, fy=400.0, cx=319.5, cy=239.5),
    "replica": dict(W=640, H=320, fx=320.0, fy=600.0 * 320.0 / 680.0, cx=599.5 * 640.0 / 1200.0, cy=339.5 * 320.0 / 680.0),
    "tiny": dict(W=96, H=64, fx=70.0, fy=70.0, cx=47.5, cy=31.5),
    # the other two datasets the reference is configured for, after the datasets.py:94-107 scaling (non-square pixels, principal
    # point off the image centre): /root/reference/configs/TUM_RGBD/tum.yaml:40-47, configs/Scannet/scannet.yaml:48-55 -- the same
    # numbers golden G1 was generated with (tests/test_golden_host.py checks them against the fixture)
    "tum": dict(W=512, H=384, fx=517.3 * 512 / 640.0, fy=516.5 * 384 / 480.0, cx=318.6 * 512 / 640.0, cy=255.3 * 384 / 480.0),
    "scannet": dict(W=320, H=240, fx=577.59 * 320 / 640.0, fy=578.73 * 240 / 480.0, cx=318.9 * 320 / 640.0, cy=242.68 * 240 / 480.0),
}
ROOM = (6.0, 3.0, 4.0)      # x (length), y (height, down), z (width): a 6 x 4 x 3 m box


def room_points(n, gen):
    """Area-uniform points on the inner faces of the box, jittered by N(0, 1 cm)."""
    lx, ly, lz = ROOM
    areas = torch.tensor([lx * lz, lx * lz, lx * ly, lx * ly, lz * ly, lz * ly], dtype=torch.float64)
    face = torch.multinomial(areas / areas.sum(), n, replacement=True, generator=gen)
    u = torch.rand(n, generator=gen, dtype=torch.float64) - 0.5
    v = torch.rand(n, generator=gen, dtype=torch.float64) - 0.5
    p = torch.zeros(n, 3, dtype=torch.float64)
    for f in range(6):
        m = face == f
        if f < 2:      # floor / ceiling: y = +-ly/2
            p[m] = torch.stack([u[m] * lx, torch.full_like(u[m], (ly / 2) * (1 if f == 0 else -1)), v[m] * lz], 1)
        elif f < 4:    # walls z = +-lz/2
            p[m] = torch.stack([u[m] * lx, v[m] * ly, torch.full_like(u[m], (lz / 2) * (1 if f == 2 else -1))], 1)
        else:          # walls x = +-lx/2
            p[m] = torch.stack([torch.full_like(u[m], (lx / 2) * (1 if f == 4 else -1)), v[m] * ly, u[m] * lz], 1)
    p += 0.01 * torch.randn(n, 3, generator=gen, dtype=torch.float64)
    return p.float()


def room_parameters(n, seed=43, knn_fn=None, device="cpu"):
    """Raw (pre-activation) Gaussian parameters of the room, SURVEY.md 8d recipe."""
    gen = torch.Generator().manual_seed(seed)
    xyz = room_points(n, gen).to(device)
    if knn_fn is None:
        from simple_knn._C import distCUDA2 as knn_fn
    d2 = torch.clamp_min(knn_fn(xyz), 1e-7) * 0.05                       # gaussian_model.py:194-203
    log_s = torch.log(torch.sqrt(d2))[:, None].repeat(1, 3).cpu()
    log_s = log_s + 0.3 * torch.randn(n, 3, generator=gen)
    rot = torch.randn(n, 4, generator=gen)
    opac = 1.5 * torch.randn(n, 1, generator=gen)
    rgb = torch.rand(n, 3, generator=gen)
    f_dc = RGB2SH(rgb)[:, None, :]
    return dict(xyz=xyz.to(device), scaling=log_s.to(device), rotation=rot.to(device), opacity=opac.to(device),
                f_dc=f_dc.contiguous().to(device))


def keyframe_order(xyz, K=160, hfov_deg=77.0, seed=43):
    """Permutation that orders the room the way a SLAM map grows (extend_from_pcd_seq appends a random subsample of each new
    keyframe's pixels, /root/reference/src/mapper.py:959-962): for keyframe k = 0..K-1 of the orbit, ~N/K not yet taken
    points inside its horizontal field of view, then whatever is left.  A 256-Gaussian segment then spans ONE keyframe's
    frustum instead of the whole room."""
    g = torch.Generator().manual_seed(seed)
    p = xyz.detach().cpu().double()
    n = p.shape[0]
    taken = torch.zeros(n, dtype=torch.bool)
    order = []
    per = max(1, n // K)
    half = math.radians(hfov_deg) * 0.5
    for k in range(K):
        phi = 2.0 * math.pi * k / K
        eye = torch.tensor([math.cos(phi), 0.0, math.sin(phi)], dtype=torch.float64)
        d = p - eye
        ang = torch.atan2(d[:, 2], d[:, 0]) - phi
        ang = torch.atan2(torch.sin(ang), torch.cos(ang)).abs()
        cand = torch.nonzero((ang < half) & ~taken).flatten()
        if cand.numel() > per:
            cand = cand[torch.randperm(cand.numel(), generator=g)[:per]]
        taken[cand] = True
        order.append(cand)
    order.append(torch.nonzero(~taken).flatten())
    return torch.cat(order)


def orbit_w2c(k, K, radius=1.0):
    """Camera k of K on a circle of radius 1 m at mid height, looking outward, yaw sweeping 360 degrees."""
    phi = 2.0 * math.pi * k / K
    fwd = np.array([math.cos(phi), 0.0, math.sin(phi)])
    down = np.array([0.0, 1.0, 0.0])
    right = np.cross(down, fwd)
    R = np.stack([right, down, fwd])                 # rows: camera x, y, z axes in world coordinates
    eye = radius * fwd
    w2c = np.eye(4)
    w2c[:3, :3] = R
    w2c[:3, 3] = -R @ eye
    return torch.from_numpy(w2c).float()


def model_from_parameters(params, config=None, device="cuda:0", knn_fn=None):
    config = config or DEFAULT_CONFIG
    gm = GaussianModel(0, config=config, device=device, knn_fn=knn_fn)
    gm.init_lr(6.0)
    gm.training_setup(OptParams(config["mapping"].get("opt_params")))
    n = params["xyz"].shape[0]
    f_rest = torch.zeros((n, 0, 3), device=device)
    P = lambda t: torch.nn.Parameter(t.detach().to(device).float().clone().contiguous().requires_grad_(True))
    gm.densification_postfix(P(params["xyz"]), P(params["f_dc"]), P(f_rest), P(params["opacity"]), P(params["scaling"]),
                             P(params["rotation"]), new_kf_ids=torch.zeros(n).int(), new_n_obs=torch.zeros(n).int())
    return gm


def make_camera(uid, w2c, intr, color, depth, device):
    W, H = intr["W"], intr["H"]
    proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=intr["fx"], fy=intr["fy"], cx=intr["cx"], cy=intr["cy"],
                                W=W, H=H).transpose(0, 1).to(device=device)
    cam = Camera(uid, color, depth, w2c.to(device), proj, intr["fx"], intr["fy"], intr["cx"], intr["cy"],
                 focal2fov(intr["fx"], W), focal2fov(intr["fy"], H), H, W, device=device)
    cam.update_RT(w2c[:3, :3].to(device), w2c[:3, 3].to(device))       # mapper.py:939
    return cam


@torch.no_grad()
def make_views(params, K, intr, device, seed=43, perturb=True, config=None):
    """K cameras whose ground-truth colour/depth are rendered from a perturbed copy of the scene
    (xyz + N(0, 5 mm), colour + N(0, 0.05)): a non-trivial loss with realistic upstream gradients."""
    from splat_slam_amd.mapper import PipelineParams
    from splat_slam_amd.renderer import render
    gen = torch.Generator().manual_seed(seed + 1)
    tgt = {k: v.clone().cpu() for k, v in params.items()}
    if perturb:
        tgt["xyz"] = tgt["xyz"] + 0.005 * torch.randn(tgt["xyz"].shape, generator=gen)
        tgt["f_dc"] = tgt["f_dc"] + (0.05 / 0.28209479177387814) * torch.randn(tgt["f_dc"].shape, generator=gen)
    gm = model_from_parameters(tgt, config=config, device=device, knn_fn=lambda p: torch.ones(p.shape[0], device=p.device))
    bg = torch.zeros(3, device=device)
    H, W = intr["H"], intr["W"]
    cams = []
    for k in range(K):
        w2c = orbit_w2c(k, K)
        cam = make_camera(k, w2c, intr, torch.zeros(3, H, W, device=device), torch.zeros(H, W, device=device), device)
        with torch.no_grad():                 # (ground truth: no autograd graph -- and no saved block -- kept alive by the images)
            pkg = render(cam, gm, PipelineParams(), bg)
        cam.original_image = pkg["render"].detach().clamp(0, 1).contiguous()
        cam.depth = pkg["depth"][0].detach().contiguous()
        cams.append(cam)
    return cams


@torch.no_grad()
def keyframe_stream(num_frames, intr, device, n_world=60000, seed=43, sweep_deg=80.0):
    """Mapping-only feed: (video_idx, frame_idx, color[3,H,W], depth[H,W], w2c[4,4]) for `num_frames` cameras sweeping
    `sweep_deg` of the orbit, observed from a dense, opaque ground-truth version of the room.  Stands in for the tracker
    messages + DepthVideo.get_depth_and_pose + frame_reader; poses are exact, depth is the rendered one."""
    from splat_slam_amd.mapper import PipelineParams
    from splat_slam_amd.renderer import render
    world = room_parameters(n_world, seed=seed, device=device)
    world["scaling"] = world["scaling"] * 0 + world["scaling"].mean(dim=1, keepdim=True) + 1.6    # opaque surface splats
    world["opacity"] = torch.full_like(world["opacity"], 4.0)
    gm = model_from_parameters(world, device=device, knn_fn=lambda p: torch.ones(p.shape[0], device=p.device))
    bg = torch.zeros(3, device=device)
    H, W = intr["H"], intr["W"]
    K = int(round(360.0 / sweep_deg * max(1, num_frames - 1))) if num_frames > 1 else 1
    out = []
    for k in range(num_frames):
        w2c = orbit_w2c(k, max(K, 1))
        cam = make_camera(k, w2c, intr, torch.zeros(3, H, W, device=device), torch.zeros(H, W, device=device), device)
        pkg = render(cam, gm, PipelineParams(), bg)
        opa = pkg["opacity"][0]
        depth = torch.where(opa > 0.5, pkg["depth"][0] / opa.clamp_min(1e-6), torch.zeros_like(opa))
        out.append((k, k, pkg["render"].clamp(0, 1).contiguous(), depth.contiguous(), w2c))
    return out
