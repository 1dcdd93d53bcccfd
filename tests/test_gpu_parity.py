"""-m gpu: the HIP rasterizer (through the C ABI and the drop-in Python surface) against the fp64 oracle.

Tolerance (BASELINE.json north_star): 1e-4 relative, fp32, per output / gradient tensor, measured as
max|a-b| / max|b|.  The rasterizer is piecewise continuous (alpha >= 1/255, T >= 1e-4, integer radii): a pair that
sits within one fp32 ulp of a cut-off may legitimately fall on different sides in fp32 and fp64.  Such a flip
changes one pixel by at most alpha_min = 1/255 of a colour, so images are additionally allowed a handful of
outlier pixels bounded by 1.01/255 * max|value|; everything else must meet 1e-4.
"""
import pytest
import torch

from helpers import random_scene
from gpu_utils import GRAD_KEYS, outlier_report, rel_linf, run_hip, run_oracle, to_fp32_inputs

pytestmark = pytest.mark.gpu
REL = 1e-4

CASES = [
    # name, n, W, H, kwargs
    ("tiny", 7, 32, 32, dict(fx=30.0, fy=28.0, cx=15.2, cy=16.4)),
    ("one", 1, 16, 16, dict(fx=20.0, fy=20.0, cx=8.0, cy=8.0, spread=0.2)),
    ("two", 2, 16, 16, dict(fx=20.0, fy=20.0, cx=8.0, cy=8.0, spread=0.3)),
    ("ragged", 60, 50, 37, dict(fx=44.0, fy=41.0, cx=24.1, cy=19.3)),          # H, W not multiples of 8/16
    ("narrow", 60, 16, 64, dict(fx=18.0, fy=40.0, cx=7.6, cy=31.2)),           # ONE 16-pixel super-tile column (the tile
                                                                                #  mapping's reciprocal of sgx wrapped for sgx = 1)
    ("dense", 1000, 64, 48, dict(fx=55.0, fy=52.0, cx=30.7, cy=24.9, scale_range=(0.01, 0.12))),
    ("bg", 200, 64, 48, dict(fx=55.0, fy=52.0, cx=30.7, cy=24.9, bg=torch.tensor([0.3, 0.6, 0.1]).double())),
    ("wide", 300, 96, 64, dict(fx=40.0, fy=40.0, cx=47.5, cy=31.5, spread=2.0, scale_range=(0.02, 0.6))),
    # long per-tile lists: 'skewed' keeps the light forward build (256-key LDS sort) and pushes the central tiles onto
    # the in-HBM sort path; 'heavy' selects the 4096-key build and the multi-chunk (64-lane) backward
    ("skewed", 2500, 256, 256, dict(fx=220.0, fy=220.0, cx=127.5, cy=127.5, spread=0.12, scale_range=(0.004, 0.03))),
    ("heavy", 3000, 64, 48, dict(fx=55.0, fy=52.0, cx=30.7, cy=24.9, scale_range=(0.05, 0.3))),
    ("sh1", 150, 64, 48, dict(fx=55.0, fy=52.0, cx=30.7, cy=24.9, sh_degree=1)),
    ("sh2", 150, 64, 48, dict(fx=55.0, fy=52.0, cx=30.7, cy=24.9, sh_degree=2)),
    ("sh3", 150, 64, 48, dict(fx=55.0, fy=52.0, cx=30.7, cy=24.9, sh_degree=3)),
]


def _weights(seed, H, W):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(3, H, W, generator=g, dtype=torch.float64), torch.randn(1, H, W, generator=g, dtype=torch.float64)


def _check_images(hip, ref, name):
    col, radii, dep, opa, nt = hip
    rcol, rradii, rdep, ropa, rnt = ref
    assert torch.equal(radii, rradii), f"{name}: radii differ"
    for a, b, what in ((col, rcol, "color"), (dep, rdep, "depth"), (opa, ropa, "opacity")):
        n_out, emax, m = outlier_report(a, b, REL)
        assert n_out <= max(2, a.numel() // 2000), f"{name}/{what}: {n_out} pixels beyond {REL} (max err {emax}, max {m})"
        assert emax <= 1.01 / 255.0 * max(m, 1.0), f"{name}/{what}: max err {emax}"
    # n_touched: identical unless a cut-off flipped (each flip moves one count by one)
    assert (nt - rnt).abs().sum().item() <= max(2, nt.numel() // 500), f"{name}: n_touched differs"
    assert torch.equal(nt > 0, rnt > 0) or (nt - rnt).abs().max().item() <= 1


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_forward_and_backward_match_oracle(case):
    name, n, W, H, kw = case
    inp, s = random_scene(n, seed=11, W=W, H=H, **kw)
    inp, s = to_fp32_inputs(inp, s)
    wc, wd = _weights(5, H, W)
    hip_out, hip_g = run_hip(inp, s, wc, wd)
    ref_out, ref_g = run_oracle(inp, s, wc, wd, dtype=torch.float64)
    _check_images(hip_out, ref_out, name)
    for k in GRAD_KEYS:
        assert hip_g[k] is not None, f"{name}: no gradient for {k}"
        r = rel_linf(hip_g[k].reshape(-1), ref_g[k].reshape(-1))
        assert r <= REL or ref_g[k].abs().max() == 0, f"{name}: grad {k} rel err {r}"
    assert torch.all(hip_g["means2D"][:, 2] == 0)


@pytest.mark.parametrize("mod", [0.6, 1.7])
def test_scale_modifier_matches_oracle(mod):
    """`scale_modifier` of GaussianRasterizationSettings (render(..., scaling_modifier), gaussian_renderer/__init__.py:24,63): S = diag(mod * s)
    in the covariance, forward and backward (the scale gradient carries the factor)."""
    inp, s = random_scene(300, seed=13, W=64, H=48)
    inp, s = to_fp32_inputs(inp, s)
    s = s._replace(scale_modifier=mod)
    wc, wd = _weights(6, 48, 64)
    hip_out, hip_g = run_hip(inp, s, wc, wd)
    ref_out, ref_g = run_oracle(inp, s, wc, wd, dtype=torch.float64)
    _check_images(hip_out, ref_out, f"mod{mod}")
    for k in GRAD_KEYS:
        r = rel_linf(hip_g[k].reshape(-1), ref_g[k].reshape(-1))
        assert r <= REL or ref_g[k].abs().max() == 0, f"scale_modifier {mod}: grad {k} rel err {r}"
    base_out, _ = run_hip(inp, s._replace(scale_modifier=1.0), wc, wd)
    assert not torch.equal(base_out[0], hip_out[0])          # (the factor really changes the render)


def test_precomputed_colour_and_covariance_inputs():
    from oracle import raster_oracle as O
    inp, s = random_scene(120, seed=3, W=64, H=48)
    inp, s = to_fp32_inputs(inp, s)
    R = O.quat_to_rot(inp["rotations"])
    Mx = R * inp["scales"][:, None, :]
    S = Mx @ Mx.transpose(1, 2)
    cov6 = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)
    rgb = torch.clamp_min(O.SH_C0 * inp["shs"][:, 0] + 0.5, 0.0)
    alt = dict(means3D=inp["means3D"], means2D=inp["means2D"], opacities=inp["opacities"], colors_precomp=rgb.float().double(),
               cov3D_precomp=cov6.float().double(), theta=inp["theta"], rho=inp["rho"])
    wc, wd = _weights(6, 48, 64)
    hip_out, hip_g = run_hip(alt, s, wc, wd)
    ref_out, ref_g = run_oracle(alt, s, wc, wd)
    _check_images(hip_out, ref_out, "precomp")
    for k in ["means3D", "means2D", "opacities", "colors_precomp", "cov3D_precomp", "theta", "rho"]:
        assert rel_linf(hip_g[k].reshape(-1), ref_g[k].reshape(-1)) <= REL, k


def test_nothing_visible_renders_background():
    inp, s = random_scene(5, seed=1, W=32, H=32, bg=torch.tensor([0.1, 0.2, 0.3]).double())
    inp["means3D"] = inp["means3D"] * 0 + torch.tensor([0.0, 0.0, -50.0]).double()    # far behind any of our cameras
    wc, wd = _weights(1, 32, 32)
    hip_out, hip_g = run_hip(inp, s, wc, wd)
    ref_out, _ = run_oracle(inp, s)
    if int((ref_out[1] > 0).sum()) == 0:
        assert int((hip_out[1] > 0).sum()) == 0
        assert torch.allclose(hip_out[0], torch.tensor([0.1, 0.2, 0.3]).view(3, 1, 1).expand(3, 32, 32))
        assert hip_out[2].abs().max() == 0 and hip_out[3].abs().max() == 0 and hip_out[4].abs().max() == 0
        for k in GRAD_KEYS:
            assert hip_g[k].abs().max() == 0


def test_backward_is_bitwise_deterministic():
    inp, s = random_scene(800, seed=4, W=64, H=48)
    wc, wd = _weights(2, 48, 64)
    a_out, a_g = run_hip(inp, s, wc, wd)
    b_out, b_g = run_hip(inp, s, wc, wd)
    for x, y in zip(a_out, b_out):
        assert torch.equal(x, y)
    for k in GRAD_KEYS:
        assert torch.equal(a_g[k], b_g[k]), k


def test_forward_only_then_drop_graph_and_cpu_tensor_raises():
    from diff_gaussian_rasterization import GaussianRasterizer
    from gpu_utils import hip_settings
    inp, s = random_scene(50, seed=8, W=32, H=32)
    out, _ = run_hip(inp, s)          # forward with grad-capable path but no backward (mapper.py:972)
    assert out[0].shape == (3, 32, 32)
    rast = GaussianRasterizer(raster_settings=hip_settings(s, "cuda:0"))
    with pytest.raises(RuntimeError):
        rast(means3D=inp["means3D"].float(), means2D=inp["means2D"].float(), shs=inp["shs"].float(),
             opacities=inp["opacities"].float(), scales=inp["scales"].float(), rotations=inp["rotations"].float())


def test_upstream_pose_jacobian_switch_on_both_sides():
    """SGR_OPT_UPSTREAM_POSE_JACOBIAN / oracle UPSTREAM_POSE_JACOBIAN: with an off-centre principal point
    the pose gradient changes -- identically on both sides --, every other gradient is bit-identical to the default."""
    from oracle import raster_oracle as O
    from splat_slam_amd import _native as nat
    lib = nat.lib()
    inp, s = random_scene(300, seed=11, W=50, H=37, fx=44.0, fy=41.0, cx=19.3, cy=23.9)       # |P02|, |P12| ~ 0.2-0.3
    inp, s = to_fp32_inputs(inp, s)
    wc, wd = _weights(5, 37, 50)
    _, g_off = run_hip(inp, s, wc, wd)
    try:
        lib.sgr_set_option(nat.SGR_OPT_UPSTREAM_POSE_JACOBIAN, 1)
        O.UPSTREAM_POSE_JACOBIAN = True
        _, g_on = run_hip(inp, s, wc, wd)
        _, ref_on = run_oracle(inp, s, wc, wd)
    finally:
        lib.sgr_set_option(nat.SGR_OPT_UPSTREAM_POSE_JACOBIAN, 0)
        O.UPSTREAM_POSE_JACOBIAN = False
    _, ref_off = run_oracle(inp, s, wc, wd)
    for k in GRAD_KEYS:
        if k in ("theta", "rho"):
            assert rel_linf(g_on[k], ref_on[k]) <= REL and rel_linf(g_off[k], ref_off[k]) <= REL, k
        else:
            assert torch.equal(g_on[k], g_off[k]), k
    assert rel_linf(g_on["rho"], g_off["rho"]) > 1e-3            # the switch really does something on this camera
