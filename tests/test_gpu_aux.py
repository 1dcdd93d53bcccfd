"""-m gpu: the small C-ABI entry points (knn, SE3, mapping loss, Adam) against their oracles / golden vectors."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "reference_vectors.npz"))


@pytest.mark.parametrize("n", [1, 2, 3, 4, 257, 5000])
def test_knn_mean_dist2_is_exact(n):
    from oracle.aux_oracle import knn_mean_dist2
    from simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(n)
    pts = torch.randn(n, 3, generator=g) * torch.tensor([3.0, 1.0, 0.2])
    if n > 10:
        pts[7] = pts[3]                      # duplicate point: distance 0 is a legal neighbour
    got = distCUDA2(pts.to(DEV)).cpu().double()
    ref = knn_mean_dist2(pts)
    assert torch.allclose(got, ref, rtol=2e-5, atol=1e-9)


def _knn_ref_chunked(pts, rows=500):
    """oracle.aux_oracle.knn_mean_dist2 without the N x N matrix in memory (same arithmetic, row blocks)."""
    x = pts.double()
    out = torch.empty(x.shape[0], dtype=torch.float64)
    for a in range(0, x.shape[0], rows):
        d2 = torch.cdist(x[a:a + rows], x) ** 2
        d2[torch.arange(d2.shape[0]), torch.arange(a, a + d2.shape[0])] = float("inf")
        out[a:a + rows] = torch.topk(d2, 3, dim=1, largest=False).values.sum(dim=1) / 3.0
    return out


@pytest.mark.parametrize("shape", ["blob", "room", "plane", "outliers"])
def test_knn_grid_path_is_exact(shape):
    """From 16384 points on distCUDA2 searches a uniform grid instead of all pairs: same exact 3-NN, whatever the layout --
    anisotropic blob, surface-like room, exactly planar set (a degenerate bounding box), far outliers and duplicates (the
    search has to widen its box until the proof holds)."""
    from oracle.aux_oracle import knn_mean_dist2
    from simple_knn._C import distCUDA2
    from splat_slam_amd import synthetic as syn
    n = 40000
    g = torch.Generator().manual_seed(3)
    if shape == "room":
        pts = syn.room_points(n, g)
    else:
        pts = torch.randn(n, 3, generator=g) * torch.tensor([3.0, 1.0, 0.2])
    if shape == "plane":
        pts[:, 2] = 0.75
    if shape == "outliers":
        pts[:5] += torch.tensor([400.0, -250.0, 90.0])          # a far cluster of five
        pts[5] = torch.tensor([-1000.0, 0.0, 0.0])              # and a loner
    pts[11] = pts[4]
    pts[12] = pts[4]                                            # a triple point: two zero distances
    small = pts[:300]
    assert torch.allclose(_knn_ref_chunked(small), knn_mean_dist2(small))          # the chunked reference is the oracle
    got = distCUDA2(pts.to(DEV)).cpu().double()
    ref = _knn_ref_chunked(pts)
    assert torch.allclose(got, ref, rtol=2e-5, atol=1e-9), float((got - ref).abs().max())


def test_se3_ops_match_oracle_and_reference_exp():
    import lietorch
    from oracle import aux_oracle as A
    tau = torch.from_numpy(G["g2_tau"]).float()
    X = lietorch.SE3.exp(tau.to(DEV))
    M = X.matrix().cpu()
    assert torch.allclose(M, torch.from_numpy(G["g2_T"]), atol=2e-6)          # reference SE3_exp (pose_utils.py:66-78)
    assert torch.allclose(M.double(), A.se3_exp_matrix(tau.double()), atol=2e-6)
    assert torch.allclose(X.log().cpu(), tau, atol=3e-6)
    Minv = X.inv().matrix().cpu().double()
    assert torch.allclose(Minv, torch.linalg.inv(M.double()), atol=5e-6)      # depth_video.py:327-330 pattern
    g = torch.Generator().manual_seed(1)
    Y = lietorch.SE3.exp((0.7 * torch.randn(32, 6, generator=g)).to(DEV))
    assert torch.allclose((X * Y).matrix().cpu().double(), M.double() @ Y.matrix().cpu().double(), atol=1e-5)
    pts = torch.randn(32, 3, generator=g)
    ref = (M.double()[:, :3, :3] @ pts.double()[..., None])[..., 0] + M.double()[:, :3, 3]
    assert torch.allclose(X.act(pts.to(DEV)).cpu().double(), ref, atol=1e-5)
    a6 = torch.randn(32, 6, generator=g)
    assert torch.allclose(X.adjT(a6.to(DEV)).cpu().double(), A.adjT(M.double(), a6.double()), atol=2e-5)
    I = lietorch.SE3.Identity(4, device=DEV)
    assert torch.allclose(I.matrix().cpu(), torch.eye(4).repeat(4, 1, 1))
    dx = 0.1 * torch.randn(32, 6, generator=g)
    assert torch.allclose(X.retr(dx.to(DEV)).matrix().cpu().double(), A.se3_exp_matrix(dx.double()) @ M.double(), atol=1e-5)


@pytest.mark.parametrize("tag,init", [("g5a", False), ("g5b", True)])
def test_mapping_loss_kernel_matches_reference_golden(tag, init):
    from splat_slam_amd import _native as nat
    lib = nat.lib()
    t = lambda k: torch.from_numpy(np.asarray(G[f"{tag}_{k}"])).float().to(DEV).contiguous()
    image, depth, gt, gtd = t("image"), t("depth"), t("gt"), t("gtd")
    H, W = gtd.shape
    a = torch.tensor([0.07], device=DEV)
    b = torch.tensor([-0.03], device=DEV)
    loss, da, db = (torch.zeros(1, device=DEV) for _ in range(3))
    dimg, ddep = torch.empty_like(image), torch.empty_like(depth)
    scratch = torch.empty(1024 * 16, dtype=torch.uint8, device=DEV)
    nat.check(lib.sgr_mapping_loss(H, W, image.data_ptr(), depth.data_ptr(), gt.data_ptr(), gtd.data_ptr(),
                                   None if init else a.data_ptr(), None if init else b.data_ptr(), 0.8, 0.01, 1.0,
                                   loss.data_ptr(), dimg.data_ptr(), ddep.data_ptr(), da.data_ptr(), db.data_ptr(),
                                   scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream), "loss")
    torch.cuda.synchronize()
    assert abs(loss.item() - float(G[f"{tag}_loss"])) < 2e-7
    assert torch.allclose(dimg.cpu(), torch.from_numpy(G[f"{tag}_dimage"]), atol=1e-9)
    assert torch.allclose(ddep.cpu(), torch.from_numpy(G[f"{tag}_ddepth"]), atol=1e-9)
    if not init:
        assert abs(da.item() - float(G[f"{tag}_da"])) < 2e-6 and abs(db.item() - float(G[f"{tag}_db"])) < 2e-6


def test_adam_kernels_match_torch():
    from splat_slam_amd import _native as nat
    lib = nat.lib()
    g = torch.Generator().manual_seed(0)
    n = 10007
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=3e-3, eps=1e-15)
    p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 5):
        grad = torch.randn(n, generator=g) * 10 ** float(torch.randint(-6, 1, (1,), generator=g))
        ref.grad = grad.clone()
        opt.step()
        gd = grad.to(DEV)
        nat.check(lib.sgr_adam_step(n, p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), 3e-3, 0.9, 0.999, 1e-15, step,
                                    torch.cuda.current_stream().cuda_stream), "adam")
        assert (p.cpu() - ref.detach()).abs().max().item() < 1e-6
    # masked slab variant: only active rows move, each with its own step counter
    rows = 6
    P, Gd = torch.zeros(rows, 2, device=DEV), torch.ones(rows, 2, device=DEV)
    M, V = torch.zeros_like(P), torch.zeros_like(P)
    st = torch.zeros(rows, dtype=torch.int32, device=DEV)
    act = torch.tensor([1, 0, 1, 1, 0, 0], dtype=torch.int32, device=DEV)
    for _ in range(2):
        nat.check(lib.sgr_masked_adam(rows, 2, P.data_ptr(), Gd.data_ptr(), M.data_ptr(), V.data_ptr(), st.data_ptr(),
                                      act.data_ptr(), 0.01, 0.9, 0.999, 1e-8, torch.cuda.current_stream().cuda_stream), "madam")
    r = torch.nn.Parameter(torch.zeros(2))
    o = torch.optim.Adam([r], lr=0.01)
    for _ in range(2):
        r.grad = torch.ones(2)
        o.step()
    assert st.cpu().tolist() == [2, 0, 2, 2, 0, 0]
    assert torch.allclose(P.cpu()[[0, 2, 3]], r.detach().expand(3, 2), atol=1e-7) and P.cpu()[[1, 4, 5]].abs().max() == 0


def test_mapping_session_maps_a_keyframe_stream_and_both_loops_agree():
    import copy
    from splat_slam_amd import synthetic as syn
    from splat_slam_amd.fused import FusedMappingLoop
    from splat_slam_amd.mapper import MappingLoop
    from splat_slam_amd.session import MappingSession
    cfg = copy.deepcopy(syn.DEFAULT_CONFIG)
    tr = cfg["mapping"]["Training"]
    tr["init_itr_num"], tr["mapping_itr_num"], tr["window_size"] = 120, 12, 4
    tr["init_gaussian_update"], tr["init_gaussian_reset"] = 40, 10 ** 9
    cfg["mapping"]["opt_params"]["densify_from_iter"] = 10 ** 9
    intr = syn.INTRINSICS["tiny"]
    frames = syn.keyframe_stream(7, intr, DEV, n_world=20000, seed=5, sweep_deg=70.0)
    psnr = {}
    for name, cls in (("fused", FusedMappingLoop), ("autograd", MappingLoop)):
        torch.manual_seed(43)
        np.random.seed(43)
        sess = MappingSession(cls(cfg, device=DEV), intr)
        status = [sess.process(*f) for f in frames]
        assert status[0] == "init" and "mapped" in status[1:]
        n_mid = sess.loop.gaussians.get_xyz.shape[0]
        assert n_mid > 300
        scores = sess.finish(refine_iters=40)
        assert len(scores) == len(sess.loop.viewpoints) and all(np.isfinite(scores))
        psnr[name] = float(np.mean(scores))
    assert psnr["fused"] > 17.0 and psnr["autograd"] > 17.0, psnr
    assert abs(psnr["fused"] - psnr["autograd"]) < 2.0, psnr


def _random_model(n, seed=0):
    from splat_slam_amd.gaussian_model import GaussianModel, OptParams
    g = torch.Generator().manual_seed(seed)
    gm = GaussianModel(0, config=None, device=DEV)
    gm.training_setup(OptParams())
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    gm.extend_from_pcd(r(n, 3) * 2, r(n, 3, 1), r(n, 3) * 0.3 - 3.0, r(n, 4), r(n, 1), 0)
    gm.unique_kfIDs = torch.randint(0, 4, (n,), generator=g).int().to(DEV)
    gm.n_obs = torch.randint(0, 9, (n,), generator=g).int().to(DEV)
    for grp in gm.optimizer.param_groups:          # give Adam a state with recognisable content
        p = grp["params"][0]
        gm.optimizer.state[p] = {"step": torch.tensor(3.0), "exp_avg": torch.randn(p.shape, generator=g).to(DEV),
                                 "exp_avg_sq": torch.rand(p.shape, generator=g).to(DEV)}
    gm.xyz_gradient_accum = torch.rand(n, 1, generator=g).to(DEV)
    gm.denom = torch.randint(0, 5, (n, 1), generator=g).float().to(DEV)
    gm.max_radii2D = torch.rand(n, generator=g).to(DEV) * 30
    return gm


def test_prune_points_hip_compaction_equals_torch_indexing():
    """gaussian_model.py:519-557: one keep-list scan + one gather launch over all per-Gaussian tensors is bit-identical to
    the reference's chain of boolean-index selects (parameters, Adam moments, statistics, keyframe ids)."""
    for n, frac in ((1, 0.0), (777, 0.3), (5000, 0.95), (300, 1.0)):
        mask = (torch.rand(n, generator=torch.Generator().manual_seed(n)) < frac).to(DEV)
        a, b = _random_model(n, seed=n), _random_model(n, seed=n)
        b.use_hip_compaction = False
        a.prune_points(mask)
        b.prune_points(mask)
        assert a.get_xyz.shape[0] == int((~mask).sum())
        for name in ["_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "xyz_gradient_accum", "denom",
                     "max_radii2D", "unique_kfIDs", "n_obs"]:
            x, y = getattr(a, name), getattr(b, name)
            assert x.shape == y.shape and x.dtype == y.dtype and torch.equal(x.detach(), y.detach()), name
        for ga, gb in zip(a.optimizer.param_groups, b.optimizer.param_groups):
            sa, sb = a.optimizer.state[ga["params"][0]], b.optimizer.state[gb["params"][0]]
            assert ga["params"][0].requires_grad and ga["params"][0] is getattr(a, {"xyz": "_xyz", "f_dc": "_features_dc",
                   "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation"}[ga["name"]])
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
            assert float(sa["step"]) == 3.0


def test_map_deformation_kernel_matches_the_torch_formulation():
    """mapper.py:154-255: one pass of sgr_deform_points == the reference's torch op chain (rigid and depth-rescaled)."""
    from splat_slam_amd import deform
    from oracle.aux_oracle import se3_exp_matrix
    H, W = 40, 56
    K = torch.tensor([[50.0, 0, 27.5], [0, 48.0, 19.5], [0, 0, 1]], device=DEV)
    g = torch.Generator().manual_seed(9)
    w2c_old = torch.eye(4)
    w2c_old[:3, 3] = torch.tensor([0.1, -0.2, 4.0])
    w2c_new = (se3_exp_matrix(torch.tensor([0.05, -0.03, 0.08, 0.02, -0.04, 0.03], dtype=torch.float64)).float() @ w2c_old)
    d_old = torch.rand(H, W, generator=g) * 2 + 3
    d_new = d_old + torch.randn(H, W, generator=g) * 0.2
    d_new[::7, ::5] = 0.0                                   # invalid depths leave the scale alone
    for method in (None, "rigid"):
        res = []
        for use_hip in (True, False):
            gm = _random_model(900, seed=4)
            deform.USE_HIP = use_hip
            try:
                deform.update_mapping_points(gm, 2, w2c_new.to(DEV), w2c_old.to(DEV), d_new.to(DEV), d_old.to(DEV), K, method=method)
            finally:
                deform.USE_HIP = True
            res.append(gm)
        a, b = res
        moved = (b.unique_kfIDs == 2)
        assert int(moved.sum()) > 50
        for name, tol in (("_xyz", 2e-5), ("_rotation", 2e-6), ("_scaling", 2e-5)):
            x, y = getattr(a, name).detach(), getattr(b, name).detach()
            assert (x - y).abs().max().item() <= tol * max(1.0, y.abs().max().item()), (method, name, (x - y).abs().max().item())
        # rotations leave normalised for every Gaussian; the reference's Adam-state reset (zeros) is kept
        assert torch.allclose(a._rotation.detach().norm(dim=1), torch.ones(900, device=DEV), atol=1e-5)
        for name in ["xyz", "rotation"] + ([] if method == "rigid" else ["scaling"]):
            p = {"xyz": a._xyz, "rotation": a._rotation, "scaling": a._scaling}[name]
            assert float(a.optimizer.state[p]["exp_avg"].abs().max()) == 0.0
        assert not torch.equal(a._xyz.detach()[moved], _random_model(900, seed=4)._xyz.detach()[moved])


def test_densify_and_prune_with_hip_row_selection_equals_torch():
    """densify_and_clone's row selection through sgr_keep_list / sgr_gather_rows: the whole densify_and_prune (clone, split with
    the same torch.normal draws, prune) ends in bit-identical models."""
    res = []
    for hip in (True, False):
        gm = _random_model(1500, seed=21)
        gm.use_hip_compaction = hip
        with torch.no_grad():
            gm._scaling.data = gm._scaling.data * 0.2 - 2.5          # some above, some below the 1 % extent split threshold
        gm.xyz_gradient_accum = torch.rand(1500, 1, generator=torch.Generator().manual_seed(1)).to(DEV) * 1e-3
        gm.denom = torch.ones(1500, 1, device=DEV)
        torch.manual_seed(77)
        with torch.no_grad():
            gm.densify_and_prune(0.0002, 0.3, 6.0, 20)
        res.append(gm)
    a, b = res
    assert a.get_xyz.shape[0] == b.get_xyz.shape[0] and a.get_xyz.shape[0] != 1500
    for name in ["_xyz", "_features_dc", "_opacity", "_scaling", "_rotation", "unique_kfIDs", "n_obs", "max_radii2D", "denom"]:
        assert torch.equal(getattr(a, name).detach(), getattr(b, name).detach()), name
    for ga, gb in zip(a.optimizer.param_groups, b.optimizer.param_groups):
        assert torch.equal(a.optimizer.state[ga["params"][0]]["exp_avg"], b.optimizer.state[gb["params"][0]]["exp_avg"])


def test_config1_session_script_smoke(tmp_path, monkeypatch, capsys):
    """scripts/run_session_config1.py (the configs[1] session: default hyper-parameters, densification on, deformation of
    moved keyframes, final refinement, PSNR of the HIP map rendered by HIP and by the oracle) at smoke size."""
    import importlib.util
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_session_config1", os.path.join(root, "scripts", "run_session_config1.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / "s.json"
    monkeypatch.setattr(sys, "argv", ["x", "--keyframes", "9", "--camera", "tiny", "--refine", "60", "--oracle-views", "2",
                                      "--world", "30000", "--moved-every", "3", "--warmup-frames", "4", "--out", str(out)])
    from splat_slam_amd.gaussian_model import GaussianModel
    saved = {n: getattr(GaussianModel, n) for n in ("densify_and_prune", "extend_from_pcd_seq", "reset_opacity", "reset_opacity_nonvisible")}
    try:
        mod.main()
    finally:
        for n, f in saved.items():
            setattr(GaussianModel, n, f)
    d = json.load(open(out))
    assert d["keyframes_mapped"] >= 3 and d["gaussians_final"] > 300 and d["overflow_events"] == 0
    assert d["psnr_all_keyframes_mean"] > 15.0
    for row in d["psnr_hip_vs_oracle_render_of_the_same_map"]:
        assert abs(row["psnr_hip_map_hip_render"] - row["psnr_hip_map_oracle_render"]) < 0.05, row
