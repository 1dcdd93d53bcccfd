"""Harness parity (CPU): this repo's MappingLoop / GaussianModel / render / loss / Camera against the REFERENCE's own
loops (tests/golden/reference_loop.npz, produced by tests/golden/make_golden_loop.py running Mapper.initialize_map /
Mapper.map of /root/reference unmodified).  Both sides use the same oracle rasterizer injected under the module name
`diff_gaussian_rasterization`, so what is compared is everything AROUND the rasterizer: loop order and quirks, Adam
groups and learning rates, densification statistics, clone / split / prune, opacity resets, RNG consumption.
"""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
G = np.load(os.path.join(HERE, "golden", "reference_loop.npz"))


@pytest.fixture
def oracle_as_rasterizer(monkeypatch):
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
    monkeypatch.setitem(sys.modules, "diff_gaussian_rasterization", mod)
    for m in ("splat_slam_amd.renderer", "splat_slam_amd.mapper"):
        monkeypatch.delitem(sys.modules, m, raising=False)
    yield
    for m in ("splat_slam_amd.renderer", "splat_slam_amd.mapper"):
        sys.modules.pop(m, None)


def _close(a, b, tol, what):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert err <= tol, (what, err)


def test_loops_reproduce_the_reference_trajectory(oracle_as_rasterizer):
    from make_golden_loop import CONFIG, CX, CY, FX, FY, H, HYPER, W, scenario
    import copy
    mapper = importlib.import_module("splat_slam_amd.mapper")
    from splat_slam_amd.camera import Camera, focal2fov, getProjectionMatrix2
    cfg = copy.deepcopy(CONFIG)
    tr = cfg["mapping"]["Training"]
    tr.update(dict(init_itr_num=HYPER["init_itr_num"], init_gaussian_update=HYPER["init_gaussian_update"],
                   init_gaussian_reset=HYPER["init_gaussian_reset"], init_gaussian_th=HYPER["init_gaussian_th"],
                   init_gaussian_extent=HYPER["init_gaussian_extent"] / 6.0, mapping_itr_num=60,
                   gaussian_update_every=HYPER["gaussian_update_every"], gaussian_update_offset=HYPER["gaussian_update_offset"],
                   gaussian_th=HYPER["gaussian_th"], gaussian_extent=HYPER["gaussian_extent"] / 6.0,
                   gaussian_reset=HYPER["gaussian_reset"], size_threshold=HYPER["size_threshold"]))
    torch.manual_seed(43)
    np.random.seed(43)
    init, camdata = scenario()
    loop = mapper.MappingLoop(cfg, device="cpu", fused_loss=False)
    gm = loop.gaussians
    gm.extend_from_pcd(init["xyz"].clone(), init["feats"].clone(), init["scales"].clone(), init["rots"].clone(),
                       init["opac"].clone(), 0)
    P = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=FX, fy=FY, cx=CX, cy=CY, W=W, H=H).transpose(0, 1)
    cams = {}
    for k, color, depth, w2c in camdata:
        c = Camera(k, color, depth.numpy(), w2c, P, FX, FY, CX, CY, focal2fov(FX, W), focal2fov(FY, H), H, W, device="cpu")
        c.update_RT(c.R_gt, c.T_gt)
        cams[k] = c

    def check(tag, tol):
        assert gm._xyz.shape[0] == G[f"{tag}_xyz"].shape[0], (tag, gm._xyz.shape[0], G[f"{tag}_xyz"].shape[0])
        assert loop.iteration_count == int(G[f"{tag}_iteration_count"])
        for name, t in (("xyz", gm._xyz), ("f_dc", gm._features_dc), ("opacity", gm._opacity), ("scaling", gm._scaling),
                        ("rotation", gm._rotation), ("accum", gm.xyz_gradient_accum)):
            _close(t.detach(), G[f"{tag}_{name}"], tol, f"{tag}/{name}")
        assert torch.equal(gm.unique_kfIDs.cpu(), torch.from_numpy(G[f"{tag}_kf"]))
        assert torch.equal(gm.denom, torch.from_numpy(G[f"{tag}_denom"]))
        assert torch.equal(gm.max_radii2D, torch.from_numpy(G[f"{tag}_max_radii2D"]))
        lr = [g["lr"] for g in gm.optimizer.param_groups if g["name"] == "xyz"][0]
        assert abs(lr - float(G[f"{tag}_xyz_lr"])) < 1e-12

    # 1. map initialisation (densify at 0/5/10, opacity reset at iteration 3, no lr update: SURVEY 3.7 item 2)
    loop.viewpoints[0] = cams[0]
    loop.current_window = [0]
    loop.initialize_map(0, cams[0])
    check("init", 2e-5)
    assert torch.equal(loop.occ_aware_visibility[0], torch.from_numpy(G["init_occ0"]))
    # 2. online mapping with exposure optimisers, a densify + prune (iteration 14), a non-visible opacity reset (15)
    loop.viewpoints = {0: cams[0], 1: cams[1], 2: cams[2]}
    loop.current_window = [2, 1]
    loop.build_keyframe_optimizers()
    loop.map(loop.current_window, iters=7)
    check("map", 3e-4)      # Adam(eps=1e-15) turns 1e-7 differences (Camera builds W2C directly, the reference inverts twice) into fractions of one lr step
    exp = np.array([[cams[i].exposure_a.item(), cams[i].exposure_b.item()] for i in (0, 1, 2)])
    assert np.abs(exp - G["map_exposure"]).max() < 1e-6
    assert torch.equal(loop.occ_aware_visibility[2], torch.from_numpy(G["map_occ2"]))
    # 3. the prune pass: forward + backward, visibility refreshed, NO optimiser step
    before = gm._xyz.detach().clone()
    loop.map(loop.current_window, prune=True)
    assert torch.equal(before, gm._xyz.detach())
    assert loop.iteration_count == int(G["prune_iteration_count"])
    assert torch.equal(loop.occ_aware_visibility[1], torch.from_numpy(G["prune_occ1"]))


def test_keyframe_management_matches_reference():
    """is_keyframe / add_to_window of MappingSession vs Mapper.is_keyframe / add_to_window (mapper.py:744-831)."""
    from make_golden_loop import CX, CY, FX, FY, H, W, keyframe_cases
    from splat_slam_amd.camera import Camera, getProjectionMatrix2
    from splat_slam_amd.session import MappingSession
    P = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=FX, fy=FY, cx=CX, cy=CY, W=W, H=H).transpose(0, 1)
    for ci, case in enumerate(keyframe_cases()):
        loop = types.SimpleNamespace(config={"mapping": {"Training": {"kf_translation": 0.04, "kf_min_translation": 0.02,
                                                                       "kf_overlap": 0.95}}},
                                     device="cpu", window_size=case["window_size"])
        sess = MappingSession.__new__(MappingSession)
        sess.loop, sess.config, sess.device = loop, loop.config, "cpu"
        sess.cameras, sess.median_depth = {}, case["median_depth"]
        for k, w2c in enumerate(case["poses"]):
            c = Camera(k, None, None, w2c, P, FX, FY, CX, CY, 1.0, 1.0, H, W, device="cpu")
            c.update_RT(c.R_gt, c.T_gt)
            sess.cameras[k] = c
        occ = {k: m for k, m in enumerate(case["masks"])}
        cur = len(case["poses"]) - 1
        assert sess.is_keyframe(cur, case["window"][0], case["masks"][cur], occ) == bool(G[f"kf{ci}_is_keyframe"])
        win, removed = sess.add_to_window(cur, case["masks"][cur], occ, list(case["window"]))
        assert win == G[f"kf{ci}_window"].tolist()
        assert (-1 if removed is None else removed) == int(G[f"kf{ci}_removed"])
