#!/usr/bin/env python3
"""Generates tests/golden/reference_loop.npz by running the REFERENCE's own mapping loops on the CPU:
`Mapper.initialize_map`, `Mapper.map` (incl. the prune pass), `Mapper.final_refine` (src/mapper.py:617-710: its whole
body, numpy RNG, exposure Adam), `Mapper.update_mapping_points` (src/mapper.py:154-255: depth-rescale and rigid branches
with the Adam-moment reset), `GaussianModel` (Adam groups, densify / prune with optimiser-state surgery, opacity reset),
`render()`, `get_loss_mapping`, `Camera` -- imported unmodified from /root/reference -- against the oracle rasterizer
injected under the module name `diff_gaussian_rasterization` (SURVEY.md 8c "reference's own callers run on CPU against
an injected rasterizer").

What it pins: rows A6-A13, A9 (final_refine) and A15 (update_mapping_points) of SURVEY.md 8a (loss, loop order and
quirks, Adam hyper-parameters, learning-rate schedule, densification statistics, clone / split / prune rules, RNG
consumption, map deformation) for splat_slam_amd.mapper / gaussian_model / deform, which tests/test_loop_parity.py
replays with the same injected oracle (and tests/test_gpu_aux.py / test_gpu_fused.py on the HIP path).  It does NOT pin
the rasterizer itself.

Run in the build container only:   python tests/golden/make_golden_loop.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)


def install_stubs():
    """Modules the reference imports but this container lacks; none of their functions is called on this path."""
    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _Any(self.__name__ + "." + name)

        def __call__(self, *a, **k):
            return _Any(self.__name__ + "()")

    for name in ["cv2", "open3d", "plyfile", "colorama", "munch", "evo", "evo.core", "evo.core.trajectory", "evo.tools",
                 "evo.core.metrics", "evo.core.sync", "lietorch", "droid_backends", "torch_scatter", "simple_knn",
                 "simple_knn._C", "trimesh", "torchvision", "torchvision.transforms", "matplotlib", "matplotlib.pyplot"]:
        forced = name in ("lietorch", "simple_knn", "simple_knn._C")      # never this repo's GPU drop-ins here
        if name not in sys.modules or forced:
            try:
                if forced:
                    raise ImportError
                __import__(name)
            except Exception:
                sys.modules[name] = _Any(name)
    sys.modules["colorama"].Fore = types.SimpleNamespace(GREEN="", RED="", YELLOW="", BLUE="", CYAN="", MAGENTA="", WHITE="")
    sys.modules["colorama"].Style = types.SimpleNamespace(RESET_ALL="")
    sys.modules["munch"].munchify = lambda d: types.SimpleNamespace(**d)
    sys.modules["simple_knn._C"].distCUDA2 = lambda pts: torch.ones(pts.shape[0])
    # device="cuda" -> cpu in the torch factories the reference calls; Tensor.cuda() is the identity
    for fn in ["zeros", "ones", "empty", "tensor", "zeros_like", "ones_like", "eye", "rand", "randn", "full", "arange", "normal"]:
        orig = getattr(torch, fn)

        def wrap(*a, __orig=orig, **k):
            if "device" in k and str(k["device"]).startswith("cuda"):
                k["device"] = "cpu"
            return __orig(*a, **k)
        setattr(torch, fn, wrap)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    # the oracle under the name the reference imports (gaussian_renderer/__init__.py:15-18)
    from oracle import raster_oracle as O
    mod = types.ModuleType("diff_gaussian_rasterization")
    mod.GaussianRasterizationSettings = O.OracleSettings

    class GaussianRasterizer(torch.nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None, theta=None, rho=None):
            return O.rasterize(means3D, means2D, opacities, shs=shs, colors_precomp=colors_precomp, scales=scales,
                               rotations=rotations, cov3D_precomp=cov3D_precomp, theta=theta, rho=rho,
                               settings=self.raster_settings)
    mod.GaussianRasterizer = GaussianRasterizer
    sys.modules["diff_gaussian_rasterization"] = mod


# ---- the shared tiny scenario (also imported by tests/test_loop_parity.py)
W, H, FX, FY, CX, CY = 24, 16, 20.0, 20.0, 11.5, 7.5
CONFIG = {
    "mapping": {"BA": False, "pcd_downsample": 32, "pcd_downsample_init": 16, "adaptive_pointsize": True, "point_size": 0.05,
                "sensor_type": "depth",
                "Training": {"ssim_loss": False, "alpha": 0.8, "rgb_boundary_threshold": 0.01, "pose_window": 5,
                             "window_size": 10, "gt_camera": False, "lr": {"cam_rot_delta": 0.003, "cam_trans_delta": 0.001}},
                "opt_params": {"position_lr_init": 0.00016, "position_lr_final": 0.0000016, "position_lr_delay_mult": 0.01,
                               "position_lr_max_steps": 30000, "feature_lr": 0.0025, "opacity_lr": 0.05,
                               "scaling_lr": 0.001, "rotation_lr": 0.001, "percent_dense": 0.01, "lambda_dssim": 0.2,
                               "densify_from_iter": 10 ** 9, "densify_grad_threshold": 0.0002}}}
HYPER = dict(init_itr_num=12, init_gaussian_update=5, init_gaussian_reset=3, init_gaussian_th=0.005,
             init_gaussian_extent=180.0, gaussian_update_every=6, gaussian_update_offset=2, gaussian_th=0.0105,
             gaussian_extent=6.0, gaussian_reset=5, size_threshold=20, window_size=10)


def scenario(seed=7, n=220):
    g = torch.Generator().manual_seed(seed)
    z = 1.5 + 1.5 * torch.rand(n, generator=g)
    xyz = torch.stack([(torch.rand(n, generator=g) - 0.5) * 1.4 * z * (W / (2 * FX)),
                       (torch.rand(n, generator=g) - 0.5) * 1.4 * z * (H / (2 * FY)), z], 1)
    feats = torch.zeros(n, 3, 1)
    feats[:, :, 0] = (torch.rand(n, 3, generator=g) - 0.5) / 0.28209479177387814
    scales = torch.log(0.04 + 0.25 * torch.rand(n, 3, generator=g))
    rots = torch.nn.functional.normalize(torch.randn(n, 4, generator=g))
    opac = torch.randn(n, 1, generator=g)
    cams = []
    for k in range(3):
        ang = 0.12 * (k - 1)
        R = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
        w2c = torch.eye(4)
        w2c[:3, :3] = R
        w2c[:3, 3] = torch.tensor([0.05 * (k - 1), 0.0, 0.02 * k])
        color = torch.rand(3, H, W, generator=g)
        color[:, :2, :3] = 0.0
        depth = 1.5 + 1.5 * torch.rand(H, W, generator=g)
        depth[5:7, 10:14] = 0.0
        cams.append((k, color, depth, w2c))
    return dict(xyz=xyz, feats=feats, scales=scales, rots=rots, opac=opac), cams


def deformation_cases(seed=23):
    """Two moved keyframes (mapper.py:1021-1055 calls update_mapping_points once per moved keyframe): keyframe 1 through
    the depth-rescale branch -- with zero depths (rigid fallback per point, :224-226) and a depth drop large enough to
    make the rescale factor negative (:227-228) --, keyframe 2 through the rigid branch (:155-182)."""
    g = torch.Generator().manual_seed(seed)

    def pose(ax, ang, t):
        c, s_ = np.cos(ang), np.sin(ang)
        R = {0: [[1, 0, 0], [0, c, -s_], [0, s_, c]], 1: [[c, 0, s_], [0, 1, 0], [-s_, 0, c]], 2: [[c, -s_, 0], [s_, c, 0], [0, 0, 1]]}[ax]
        m = torch.eye(4)
        m[:3, :3] = torch.tensor(R, dtype=torch.float32)
        m[:3, 3] = torch.tensor(t, dtype=torch.float32)
        return m
    K = torch.tensor([[FX, 0.0, CX], [0.0, FY, CY], [0.0, 0.0, 1.0]])
    d_old = 1.5 + 1.5 * torch.rand(H, W, generator=g)
    d_new = d_old + 0.2 * torch.randn(H, W, generator=g)
    d_new[2:4, 3:9] = 0.0                 # no new depth: those points only move rigidly
    d_old[10:12, 15:20] = 0.0
    d_new[6:9, 10:14] = d_old[6:9, 10:14] - 6.0      # rescale factor <= 0 -> reset to 1
    return [dict(frame_idx=1, method=None, w2c_old=pose(1, 0.0, [0.0, 0.0, 0.0]), w2c_new=pose(1, 0.07, [0.03, -0.01, 0.02]),
                 depth=d_new, depth_old=d_old, K=K),
            dict(frame_idx=2, method="rigid", w2c_old=pose(0, 0.02, [0.05, 0.0, 0.04]), w2c_new=pose(2, -0.05, [0.0, 0.02, -0.03]),
                 depth=d_new, depth_old=d_old, K=K)]


def anchor_ids(n):
    """Synthetic keyframe anchors for the deformation stage (the tiny scenario seeds everything from keyframe 0)."""
    return (torch.arange(n) % 3).int()


def keyframe_cases(seed=11):
    g = torch.Generator().manual_seed(seed)
    cases = []
    for ci in range(8):
        nk = 5 + ci                                  # cameras 0..nk-1; the last one is the candidate
        poses = []
        for k in range(nk):
            w2c = torch.eye(4)
            ang = 0.05 * k + 0.02 * torch.randn((), generator=g).item()
            w2c[:3, :3] = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
            w2c[:3, 3] = torch.tensor([0.03 * k, 0.0, 0.01 * k]) + 0.01 * torch.randn(3, generator=g)
            poses.append(w2c)
        base = torch.rand(300, generator=g)
        masks = [((base + 0.25 * torch.rand(300, generator=g) * (1 + 0.3 * abs(k - nk + 1))) > 0.55).long() for k in range(nk)]
        window = list(range(nk - 2, -1, -1))          # newest first, like current_window
        cases.append(dict(poses=poses, masks=masks, window=window, median_depth=1.0 + 0.5 * ci,
                          window_size=4 + (ci % 3) * 3))
    return cases


def snapshot(gm):
    return {"xyz": gm._xyz.detach().clone(), "f_dc": gm._features_dc.detach().clone(), "opacity": gm._opacity.detach().clone(),
            "scaling": gm._scaling.detach().clone(), "rotation": gm._rotation.detach().clone(),
            "kf": gm.unique_kfIDs.clone(), "max_radii2D": gm.max_radii2D.clone(), "accum": gm.xyz_gradient_accum.clone(),
            "denom": gm.denom.clone()}


def main():
    install_stubs()
    from src.mapper import Mapper
    from thirdparty.gaussian_splatting.scene.gaussian_model import GaussianModel
    from thirdparty.gaussian_splatting.utils.graphics_utils import focal2fov, getProjectionMatrix2
    from thirdparty.monogs.utils.camera_utils import Camera
    import torch as T
    T.autograd.set_detect_anomaly(False)
    T.set_num_threads(1)        # single-threaded reductions: the replay (same setting) is then comparable bit for bit on any host
    T.manual_seed(43)
    np.random.seed(43)
    init, camdata = scenario()
    fake = types.SimpleNamespace()
    fake.config = CONFIG
    fake.printer = types.SimpleNamespace(print=lambda *a, **k: None)
    fake.opt_params = types.SimpleNamespace(**CONFIG["mapping"]["opt_params"])
    fake.pipeline_params = types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    fake.background = T.zeros(3)
    fake.online_plotting = False
    fake.iteration_count, fake.last_sent = 0, 0
    fake.occ_aware_visibility, fake.viewpoints, fake.current_window = {}, {}, []
    for k, v in HYPER.items():
        setattr(fake, k, v)
    gm = GaussianModel(0, config=CONFIG)
    gm.init_lr(6.0)
    gm.training_setup(fake.opt_params)
    gm.extend_from_pcd(init["xyz"].clone(), init["feats"].clone(), init["scales"].clone(), init["rots"].clone(),
                       init["opac"].clone(), 0)
    fake.gaussians = gm
    P = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=FX, fy=FY, cx=CX, cy=CY, W=W, H=H).transpose(0, 1)
    cams = {}
    for k, color, depth, w2c in camdata:
        c = Camera(k, color, depth.numpy(), w2c, P, FX, FY, CX, CY, focal2fov(FX, W), focal2fov(FY, H), H, W, device="cpu")
        c.update_RT(c.R_gt, c.T_gt)
        cams[k] = c
    out = {}

    def put(tag):
        for k, v in snapshot(gm).items():
            out[f"{tag}_{k}"] = v.numpy()
        out[f"{tag}_iteration_count"] = np.array(fake.iteration_count)
        out[f"{tag}_xyz_lr"] = np.array([g["lr"] for g in gm.optimizer.param_groups if g["name"] == "xyz"][0])

    def put_adam(tag):
        for g in gm.optimizer.param_groups:
            st = gm.optimizer.state.get(g["params"][0])
            if st is not None and g["name"] != "f_rest":
                out[f"{tag}_m_{g['name']}"] = st["exp_avg"].numpy().copy()
                out[f"{tag}_v_{g['name']}"] = st["exp_avg_sq"].numpy().copy()
                out[f"{tag}_step_{g['name']}"] = np.array(float(st["step"]))

    # 1. initialize_map on keyframe 0 (densify at it 0,5,10; opacity reset at iteration_count 3)
    fake.viewpoints[0] = cams[0]
    fake.current_window = [0]
    Mapper.initialize_map(fake, 0, cams[0])
    put("init")
    print("N after init", gm.get_xyz.shape[0], "opacity range", float(gm.get_opacity.min()), float(gm.get_opacity.max()))
    out["init_occ0"] = fake.occ_aware_visibility[0].numpy()
    # 2. online mapping: window [2, 1, 0] with exposure optimisers like mapper.py:1067-1111, 7 iterations + prune pass
    fake.viewpoints = {0: cams[0], 1: cams[1], 2: cams[2]}
    fake.current_window = [2, 1]                      # keyframe 0 stays outside: it becomes the "random" view
    opt_params = []
    for idx in fake.current_window:
        opt_params.append({"params": [cams[idx].exposure_a], "lr": 0.01, "name": f"exposure_a_{idx}"})
        opt_params.append({"params": [cams[idx].exposure_b], "lr": 0.01, "name": f"exposure_b_{idx}"})
    fake.keyframe_optimizers = T.optim.Adam(opt_params)
    Mapper.map(fake, fake.current_window, iters=7)
    put("map")
    put_adam("map")
    # keyframe (exposure) optimiser state, rows = cameras 0..2, columns = (a, b); camera 0 is not in the optimiser
    kst = np.zeros((3, 2, 3))
    for g in fake.keyframe_optimizers.param_groups:
        st = fake.keyframe_optimizers.state[g["params"][0]]
        kind, idx = g["name"].rsplit("_", 1)
        kst[int(idx), 0 if kind == "exposure_a" else 1] = [float(st["exp_avg"]), float(st["exp_avg_sq"]), float(st["step"])]
    out["map_exposure_adam"] = kst
    out["map_exposure"] = np.array([[cams[i].exposure_a.item(), cams[i].exposure_b.item()] for i in (0, 1, 2)])
    out["map_occ2"] = fake.occ_aware_visibility[2].numpy()
    Mapper.map(fake, fake.current_window, prune=True)
    out["prune_occ1"] = fake.occ_aware_visibility[1].numpy()
    out["prune_iteration_count"] = np.array(fake.iteration_count)

    # 2b. final refinement (mapper.py:617-710, the whole method): no keyframes to re-read from the tracker (empty
    # video_idxs), then 9 iterations of ONE numpy-random view + Adam + lr update + exposure Adam of the last window
    from src.utils.Printer import FontColor  # noqa: F401  (imported by mapper.py itself)
    fake.video_idxs, fake.keyframe_idxs = [], []
    np.random.seed(1234)
    Mapper.final_refine(fake, iters=1)       # the step that also carries what the prune pass left in every .grad
    put("refine1")
    put_adam("refine1")
    out["refine1_exposure"] = np.array([[cams[i].exposure_a.item(), cams[i].exposure_b.item()] for i in (0, 1, 2)])
    Mapper.final_refine(fake, iters=8)
    put("refine")
    put_adam("refine")
    out["refine_exposure"] = np.array([[cams[i].exposure_a.item(), cams[i].exposure_b.item()] for i in (0, 1, 2)])
    # 2c. map deformation (mapper.py:154-255) for two moved keyframes, then 3 more refinement iterations so that the
    # Adam-moment reset (replace_tensor_to_optimizer) is observable in the trajectory
    fake.device = "cpu"
    gm.unique_kfIDs = anchor_ids(gm.get_xyz.shape[0])
    for ci, c in enumerate(deformation_cases()):
        Mapper.update_mapping_points(fake, c["frame_idx"], c["w2c_new"].clone(), c["w2c_old"].clone(), c["depth"].clone(),
                                     c["depth_old"].clone(), c["K"].clone(), method=c["method"])
        put(f"deform{ci}")
        put_adam(f"deform{ci}")
    np.random.seed(4321)
    Mapper.final_refine(fake, iters=3)
    put("refine2")
    # 3. keyframe management on the consumers of n_touched (mapper.py:744-831): random masks / poses, fixed seed
    kf = keyframe_cases()
    fake.config["mapping"]["Training"].update(kf_translation=0.04, kf_min_translation=0.02, kf_overlap=0.95)
    for ci, case in enumerate(kf):
        fake.cameras = {}
        for k, w2c in enumerate(case["poses"]):
            c = Camera(k, None, None, w2c, P, FX, FY, CX, CY, 1.0, 1.0, H, W, device="cpu")
            c.update_RT(c.R_gt, c.T_gt)
            fake.cameras[k] = c
        fake.median_depth = case["median_depth"]
        fake.window_size = case["window_size"]
        occ = {k: m for k, m in enumerate(case["masks"])}
        cur = len(case["poses"]) - 1
        out[f"kf{ci}_is_keyframe"] = np.array(bool(Mapper.is_keyframe(fake, cur, case["window"][0], case["masks"][cur], occ)))
        win, removed = Mapper.add_to_window(fake, cur, case["masks"][cur], occ, list(case["window"]))
        out[f"kf{ci}_window"] = np.array(win)
        out[f"kf{ci}_removed"] = np.array(-1 if removed is None else removed)
    np.savez_compressed(os.path.join(HERE, "reference_loop.npz"), **out)
    print("N after init:", out["init_xyz"].shape[0], " after map:", out["map_xyz"].shape[0], " iterations:", fake.iteration_count)


if __name__ == "__main__":
    main()
