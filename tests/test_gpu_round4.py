"""-m gpu: what round 4 added.

  * configs[4]'s LOOP at its map size: final_refine (mapper.py:617-710) over 1.5 M Gaussians -- a single process with 8 views per
    optimiser step and 2 ranks (sharing the GPU, host-staged exchange) with one view per rank: step counters exact, replicas
    bitwise, no capacity overflow, equal to the single-process form up to summation order;
  * the rasterizer workload of configs[2] / configs[4] at the reference's other resolutions: oracle parity of the batched mapping
    path at TUM 512x384 (tum.yaml:40-41) and ScanNet 320x240 (scannet.yaml:48-49) with the dataset intrinsics of golden G1
    (non-square pixels, principal point off the image centre);
  * a parity case WITHOUT the depth-key hand-over: the oracle sorts by its OWN depths (fp64 rounded to fp32), near ties are moved
    apart first -- the hand-over of the other full-size cases cannot be hiding an ordering defect;
  * `spherical_harmonics: True` (mapper.py:78,85) makes FusedMappingLoop run the autograd loop (like `ssim_loss: True`);
  * FusedAdam bumps the parameters' version counters: a cached activation cannot survive a step.
"""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PARAMS = ["_xyz", "_features_dc", "_opacity", "_scaling", "_rotation"]
GROUPS = ["xyz", "f_dc", "opacity", "scaling", "rotation"]


# ------------------------------------------------------------------------------------------------ configs[4]: the refine loop at 1.5 M
def _c4_loop(n, views=8):
    from splat_slam_amd import synthetic as syn
    from splat_slam_amd.fused import FusedMappingLoop
    intr = syn.INTRINSICS["metric"]
    params = syn.room_parameters(n, seed=43, device=DEV)
    cams = syn.make_views(params, views, intr, DEV, seed=43)
    f = FusedMappingLoop(syn.DEFAULT_CONFIG, device=DEV)
    f.gaussians = syn.model_from_parameters(params, device=DEV)
    f.viewpoints = {c.uid: c for c in cams}
    f.current_window = list(range(views))
    f.build_keyframe_optimizers()
    f.iteration_count = 50
    return f


def _state(f):
    gm = f.gaussians
    st = {g["name"]: gm.optimizer.state[g["params"][0]] for g in gm.optimizer.param_groups}
    out = {k: getattr(gm, k).detach().cpu().clone() for k in PARAMS}
    out.update({"m_" + k: st[k]["exp_avg"].detach().cpu().clone() for k in GROUPS})
    out.update({"v_" + k: st[k]["exp_avg_sq"].detach().cpu().clone() for k in GROUPS})
    out["exposure"] = f._exp.param[:8].detach().cpu().clone()
    out["steps"] = torch.tensor([float(st[k]["step"]) for k in GROUPS])
    out["overflow"] = torch.tensor([float(f.overflow_events)])
    out["iteration_count"] = torch.tensor([float(f.iteration_count)])
    return out


def _c4_worker(rank, world, port, out, n, iters):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from splat_slam_amd.parallel import Comm
    f = _c4_loop(n)
    f.set_parallel(world, rank, split_views=True, sync="zero1", comm=Comm(staged=True))
    np.random.seed(7)
    f.final_refine(iters=iters, views_per_step="world")
    f.check_overflow()
    torch.cuda.synchronize()
    out[rank] = _state(f)
    dist.destroy_process_group()


def test_configs4_final_refine_loop_at_map_size():
    """1.5 M Gaussians, 640x480.  (a) one process, 8 distinct random views per optimiser step, 8 steps (64 renders);
    (b) 2 ranks, one view per rank and step, 8 steps -- against (c) one process rendering both views of every step itself."""
    from test_gpu_round3 import _spawn
    n = 1500000
    f = _c4_loop(n)
    np.random.seed(7)
    p0 = {k: getattr(f.gaussians, k).detach().clone() for k in PARAMS}
    f.final_refine(iters=64, views_per_step=8)
    f.check_overflow()
    torch.cuda.synchronize()
    a = _state(f)
    assert a["steps"].tolist() == [8.0] * 5 and a["overflow"].item() == 0 and a["iteration_count"].item() == 50 + 64
    for k in PARAMS:
        assert torch.isfinite(a[k]).all(), k
        moved = (a[k] - p0[k].cpu()).abs().max().item()
        assert 0 < moved < 1.0, (k, moved)              # (8 Adam steps at lr <= 0.05 cannot move anything further)
    for g in GROUPS:       # Adam moments exist for every Gaussian some view saw, and only finite values
        assert torch.isfinite(a["m_" + g]).all() and torch.isfinite(a["v_" + g]).all() and (a["v_" + g] >= 0).all(), g
        assert int((a["m_" + g].reshape(n, -1).abs().sum(dim=1) > 0).sum()) > 10000, g
    del f
    torch.cuda.empty_cache()

    r0, r1 = _spawn(_c4_worker, n, 16)
    for k in r0:
        assert torch.equal(r0[k], r1[k]), ("ranks differ", k)
    assert r0["steps"].tolist() == [8.0] * 5 and r0["overflow"].item() == 0 and r0["iteration_count"].item() == 50 + 16
    f = _c4_loop(n)
    np.random.seed(7)
    f.final_refine(iters=16, views_per_step=2)
    f.check_overflow()
    torch.cuda.synchronize()
    single = _state(f)
    assert single["steps"].tolist() == [8.0] * 5
    lr = {"_xyz": 9.6e-4, "_features_dc": 2.5e-3, "_opacity": 0.05, "_scaling": 6e-3, "_rotation": 1e-3}
    for k, step in lr.items():        # same gradient sums up to summation order: a bounded few rounding-noise flips of Adam's sign
        d = (r0[k] - single[k]).abs()
        assert (d > 0.02 * step).float().mean().item() < 0.01, k
    assert (r0["exposure"] - single["exposure"]).abs().max().item() < 2e-3


# ------------------------------------------------------------------------------------------------ TUM / ScanNet resolutions
@pytest.mark.parametrize("camera,n", [("tum", 120000), ("scannet", 120000)])
def test_batched_mapping_path_matches_oracle_at_the_other_dataset_resolutions(camera, n):
    """configs[2] (TUM fr1_desk, 512x384) and configs[4] (ScanNet, 320x240) as rasterizer workloads: forward + fused loss + backward +
    gather of 3 views against the fp64 oracle (the tracker of configs[2] stays out of scope)."""
    from test_gpu_fullsize import _run_batched_case
    _run_batched_case(n, camera, 3)


# ------------------------------------------------------------------------------------------------ no depth-key hand-over
def test_autograd_api_matches_oracle_that_sorts_by_its_own_depths():
    """The full-size cases hand the oracle the HIP forward's fp32 depths as SORT KEYS (after checking them to 8 ulp), because two
    correct implementations order splats whose depths agree to an ulp differently.  Here nothing is handed over: the oracle sorts
    by its own depths; the scene (configs[0] shape) is first moved off its knife edges AND its near ties (visible Gaussians whose
    depths are within 32 fp32 ulp of each other AND whose tile rectangles overlap are pushed apart along the viewing direction),
    then held to 1e-4 with no exceptions -- an ordering defect on the HIP side would have nowhere to hide."""
    from gpu_utils import GRAD_KEYS, move_off_knife_edges_and_depth_ties, rel_linf, run_hip, run_oracle
    from test_gpu_fullsize import REL, Soft, WIDTH, _activated_inputs, _oracle_settings, _room
    n = 20000
    syn, intr, params, cams = _room(n, "replica", 1)
    gm = syn.model_from_parameters(params, device=DEV)
    inp = _activated_inputs(gm)
    s = _oracle_settings(cams[0], intr)
    soft = Soft()
    rounds, moves = move_off_knife_edges_and_depth_ties(inp, s)
    soft.check(True, f"{moves} Gaussian moves (overlapping near ties in depth pushed apart along the viewing direction) over {rounds} rounds")
    gw = torch.Generator().manual_seed(5)
    wc = torch.randn(3, intr["H"], intr["W"], generator=gw, dtype=torch.float64)
    wd = torch.randn(1, intr["H"], intr["W"], generator=gw, dtype=torch.float64)
    hip_out, hip_g = run_hip(inp, s, wc, wd)
    ref_out, ref_g = run_oracle(inp, s, wc, wd, dtype=torch.float64)            # depth_sort_key=None: the oracle's own depths
    assert int((ref_out[1] > 0).sum()) > 1000
    soft.check(torch.equal(hip_out[1].long(), ref_out[1].long()), "radii")
    for i, what in ((0, "color"), (2, "depth"), (3, "opacity")):
        r = rel_linf(hip_out[i], ref_out[i])
        soft.check(r <= REL, f"{what}: rel err {r:.3e} (no outlier pixels allowed)")
    soft.check(torch.equal(hip_out[4].long(), ref_out[4].long()), "n_touched")
    for k in GRAD_KEYS:
        r = rel_linf(hip_g[k].reshape(-1, WIDTH.get(k, 1)), ref_g[k].reshape(-1, WIDTH.get(k, 1)))
        soft.check(r <= REL, f"grad {k}: rel err {r:.3e} (every Gaussian, no list, no hand-over)")
    soft.done()


# ------------------------------------------------------------------------------------------------ spherical_harmonics: True
def test_spherical_harmonics_config_falls_back_to_the_autograd_loop():
    """`spherical_harmonics: True` (mapper.py:78,85: sh_degree 3 -> an [N, 15, 3] f_rest group; active degree stays 0 because nobody
    calls oneupSHdegree): FusedMappingLoop runs the reference's loop through autograd -- same trajectory as MappingLoop."""
    from splat_slam_amd import synthetic as syn
    from splat_slam_amd.fused import FusedMappingLoop
    from splat_slam_amd.mapper import MappingLoop
    cfg = copy.deepcopy(syn.DEFAULT_CONFIG)
    cfg["mapping"]["Training"]["spherical_harmonics"] = True
    res = []
    for cls in (FusedMappingLoop, MappingLoop):
        intr = syn.INTRINSICS["tiny"]
        params = syn.room_parameters(3000, seed=5, device=DEV)
        params["scaling"] = params["scaling"] + 1.2
        cams = syn.make_views(params, 4, intr, DEV, seed=5)
        loop = cls(cfg, device=DEV)
        gm = loop.gaussians
        assert gm.max_sh_degree == 3
        n = params["xyz"].shape[0]
        P = lambda t: torch.nn.Parameter(t.detach().to(DEV).float().clone().contiguous().requires_grad_(True))
        gm.densification_postfix(P(params["xyz"]), P(params["f_dc"]), P(torch.zeros(n, 15, 3)), P(params["opacity"]), P(params["scaling"]),
                                 P(params["rotation"]), new_kf_ids=torch.zeros(n).int(), new_n_obs=torch.zeros(n).int())
        loop.viewpoints = {c.uid: c for c in cams}
        loop.current_window = [0, 1, 2]
        loop.build_keyframe_optimizers()
        loop.iteration_count = 50
        torch.manual_seed(1)
        loop.map(loop.current_window, iters=2)
        torch.cuda.synchronize()
        res.append({k: getattr(gm, k).detach().clone() for k in PARAMS + ["_features_rest"]})
        if cls is FusedMappingLoop:
            assert loop.autograd_fallback and loop._acc is None       # no fused state was ever built
    for k in PARAMS + ["_features_rest"]:
        assert torch.equal(res[0][k], res[1][k]), k
    assert res[0]["_features_rest"].shape == (3000, 15, 3) and float(res[0]["_features_rest"].abs().max()) == 0.0   # degree 0 is active
    assert not torch.equal(res[0]["_xyz"], torch.zeros_like(res[0]["_xyz"]))


# ------------------------------------------------------------------------------------------------ FusedAdam and the activation cache
def test_fused_adam_step_invalidates_cached_activations():
    """A grad-enabled getter read between backward and step used to leave a cached activation that later renders reused AFTER
    the raw-pointer update: FusedAdam now bumps `_version` like torch.optim.Adam's in-place ops."""
    from splat_slam_amd import synthetic as syn
    params = syn.room_parameters(2000, seed=3, device=DEV)
    gm = syn.model_from_parameters(params, device=DEV)
    from splat_slam_amd.optim import FusedAdam
    assert isinstance(gm.optimizer, FusedAdam)
    s0 = gm.get_scaling                     # cached, grad-enabled
    s0.sum().backward()
    v0 = gm._scaling._version
    s1 = gm.get_scaling                     # read between backward and step: cached again
    gm.optimizer.step()
    assert gm._scaling._version > v0
    s2 = gm.get_scaling
    torch.cuda.synchronize()
    assert s2 is not s1 and torch.equal(s2.detach(), torch.exp(gm._scaling.detach()))
    assert not torch.equal(s2.detach(), s1.detach())
