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


# The loops, Adam(eps=1e-15) included, replay the reference's trajectory to the last bit when every matrix and reduction is
# computed like the reference computes it (Camera re-inverts W2C twice like graphics_utils.py:41-45); 1e-6 only leaves
# room for a different BLAS / libm build.  A sign flip of a rounding-noise gradient would show up as ~lr (1e-3).
EXACT = 1e-6


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
    monkey_threads = torch.get_num_threads()
    torch.set_num_threads(1)          # like the generator: reductions are then independent of the host's core count
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
    check("init", EXACT)
    assert torch.equal(loop.occ_aware_visibility[0], torch.from_numpy(G["init_occ0"]))
    # 2. online mapping with exposure optimisers, a densify + prune (iteration 14), a non-visible opacity reset (15)
    loop.viewpoints = {0: cams[0], 1: cams[1], 2: cams[2]}
    loop.current_window = [2, 1]
    loop.build_keyframe_optimizers()
    loop.map(loop.current_window, iters=7)
    check("map", EXACT)
    exp = np.array([[cams[i].exposure_a.item(), cams[i].exposure_b.item()] for i in (0, 1, 2)])
    assert np.abs(exp - G["map_exposure"]).max() < 1e-6
    assert torch.equal(loop.occ_aware_visibility[2], torch.from_numpy(G["map_occ2"]))
    # 3. the prune pass: forward + backward, visibility refreshed, NO optimiser step
    before = gm._xyz.detach().clone()
    loop.map(loop.current_window, prune=True)
    assert torch.equal(before, gm._xyz.detach())
    assert loop.iteration_count == int(G["prune_iteration_count"])
    assert torch.equal(loop.occ_aware_visibility[1], torch.from_numpy(G["prune_occ1"]))

    def check_adam(tag, tol):
        for g in gm.optimizer.param_groups:
            if g["name"] == "f_rest":
                continue
            st = gm.optimizer.state[g["params"][0]]
            _close(st["exp_avg"], G[f"{tag}_m_{g['name']}"], tol, f"{tag}/exp_avg/{g['name']}")
            _close(st["exp_avg_sq"], G[f"{tag}_v_{g['name']}"], tol * tol, f"{tag}/exp_avg_sq/{g['name']}")
            assert float(st["step"]) == float(G[f"{tag}_step_{g['name']}"]), (tag, g["name"])

    # 4. final refinement (mapper.py:617-710): one numpy-random view per step, Adam, lr schedule, exposure Adam
    np.random.seed(1234)
    loop.final_refine(iters=1)      # includes the gradients the prune pass above left on every parameter (no zero_grad there)
    check("refine1", EXACT)
    check_adam("refine1", EXACT)
    loop.final_refine(iters=8)
    check("refine", EXACT)
    check_adam("refine", EXACT)
    exp = np.array([[cams[i].exposure_a.item(), cams[i].exposure_b.item()] for i in (0, 1, 2)])
    assert np.abs(exp - G["refine_exposure"]).max() < 2e-6
    assert exp[0].tolist() == [0.0, 0.0]             # keyframe 0 is not in the keyframe optimiser: never stepped
    # 5. map deformation (mapper.py:154-255): depth-rescale branch for keyframe 1, rigid branch for keyframe 2
    from make_golden_loop import anchor_ids, deformation_cases
    from splat_slam_amd.deform import update_mapping_points
    gm.unique_kfIDs = anchor_ids(gm.get_xyz.shape[0])
    for ci, c in enumerate(deformation_cases()):
        before = {k: getattr(gm, k).detach().clone() for k in ("_xyz", "_rotation", "_scaling")}
        update_mapping_points(gm, c["frame_idx"], c["w2c_new"], c["w2c_old"], c["depth"], c["depth_old"], c["K"], method=c["method"])
        moved = gm.unique_kfIDs == c["frame_idx"]
        assert torch.equal(gm._xyz.detach()[~moved], before["_xyz"][~moved])
        assert (gm._xyz.detach()[moved] - before["_xyz"][moved]).abs().max() > 1e-2
        if c["method"] is None:      # the case is built to hit the three per-point outcomes of :218-228
            ds = (gm._scaling.detach() - before["_scaling"])[moved][:, 0]
            assert int((ds == 0).sum()) >= 5 and int((ds != 0).sum()) >= 50
        else:
            assert torch.equal(gm._scaling.detach(), before["_scaling"])
        check(f"deform{ci}", 2e-6)     # (the torch formulation here composes the same maps in another order)
        check_adam(f"deform{ci}", 1e-6)
        for name in ("xyz", "rotation") + (("scaling",) if c["method"] is None else ()):
            st = gm.optimizer.state[[g for g in gm.optimizer.param_groups if g["name"] == name][0]["params"][0]]
            assert st["exp_avg"].abs().max() == 0 and st["exp_avg_sq"].abs().max() == 0       # moments reset, step kept
    np.random.seed(4321)
    loop.final_refine(iters=3)
    check("refine2", 5e-6)
    torch.set_num_threads(monkey_threads)


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
