"""-m gpu: oracle parity AT THE SIZES BASELINE.json's configs name, on the code paths bench.py times.

  configs[0] shape  room, 20 000 Gaussians, 640x320 (Replica intrinsics after datasets.py:94-104)
  configs[1] shape  room, 300 000 Gaussians, 640x480 (the metric's resolution), bench camera
  opaque            a map with log-scale + 1.6 (a converged, surface-covering map; 150 k Gaussians so that the fp64 oracle
                    stays within a minute per view): per-tile lists beyond 64 / 256 entries, LDS / in-HBM sorts, the
                    multi-chunk 64-lane backward

Two code paths are pinned against the fp64 oracle (oracle/raster_oracle.py):

  (a) the drop-in autograd API (GaussianRasterizer -> sgr_forward / sgr_backward, float pixel gradients);
  (b) the batched mapping path bench.py times: sgr_map_views over >= 4 views with the mapping loss fused into the
      compositing epilogue, pixel gradients handed over as one code byte per pixel (blend_bwd_kernel<true>),
      preprocess_bwd_dense and the gather pass -- against  sum over views of oracle-autograd through the reference's
      mapping loss (/root/reference/thirdparty/monogs/utils/slam_utils.py:71-105, pinned by golden G5).

Tolerance: 1e-4 relative (max|a-b| / max|b| per tensor), BASELINE.json north_star.  Three knife-edge effects are handled
explicitly instead of by a looser tolerance:
  * a (pixel, splat) pair is composited iff alpha >= 1/255 (and T' >= 1e-4).  Pixel centres carry ~1e-4 px of fp32 error at
    coordinate ~500, which the exponent turns into up to ~1e-4 relative in alpha: a pair whose exact alpha is within that of
    1/255 is composited by one correct implementation and skipped by another, and the Gaussian's gradients move by that
    pixel's whole contribution (measured at configs[0]: 2 such pairs in the view, 3e-3 of the largest gradient; every other
    Gaussian agrees to 1e-5).  The oracle reports the Gaussians that have such a pair (raster_oracle.knife_edge_gaussians,
    band 5e-4).  The two headline configs are first MOVED OFF their knife edges (opacities of those Gaussians nudged by
    0.2-2 %, 2 rounds) and then compared with no exception at all; the batched cases hold every Gaussian that is
    not on the list to 1e-4 and bound the listed ones (2e-2, and only a few may exceed 1e-4); the LONG-LIST cases (opaque maps:
    the list holds practically every Gaussian there and pins nothing) hold EVERY visible Gaussian to 1e-4 except a counted
    <= 0.1 % of them, bounded by 2e-3 (_per_gaussian_bounded; rounds 1-3 accepted "rel L2 <= 2e-4, rel max <= 5e-3" here);
  * splats of a tile are ordered by the fp32 BIT PATTERN of their view-space depth.  With 300 k Gaussians on planar walls
    a handful of overlapping pairs per view have depths equal to the last ulp; which of the two comes first is decided by
    the rounding of the depth itself (first full-size run: 17 pixels / 3e-3, gradients off by up to 7e-3 of the maximum
    for exactly those Gaussians).  The oracle therefore takes the HIP forward's fp32 depths as its SORT KEY only
    (sgr_query_depth_keys), after the test has checked that they equal its own depths to <= 8 ulp;
  * radii = ceil(3 sqrt(lambda)): with ~19 k visible Gaussians per view a handful sit within fp32 rounding of an
    integer; at most 3 may differ, by one (their extra / missing ring of pixels has alpha < 1/255: no image effect);
  * the L1 loss gradient is sign(residual): where the fp64 residual is smaller than 1e-5 the sign is decided by fp32
    rounding.  The test reads the code bytes the HIP epilogue wrote, REQUIRES them to equal the oracle's signs
    everywhere else, and uses the HIP sign on those knife-edge values when it differentiates the oracle.
"""
import math

import pytest
import torch

from gpu_utils import (GRAD_KEYS, check_depth_keys, hip_depth_keys, knife_ids, move_off_knife_edges, move_off_knife_edges_and_depth_ties,
                       outlier_report, rel_linf, run_hip, run_oracle)
from oracle import raster_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REL = 1e-4
KNIFE = 1e-5


class Soft:
    """Collects every violated bound of a case before failing, so one GPU run reports all of them."""

    def __init__(self):
        self.bad, self.log = [], []

    def check(self, ok, msg):
        self.log.append(("ok   " if ok else "FAIL ") + msg)
        if not ok:
            self.bad.append(msg)

    def done(self):
        print("\n".join(self.log))
        assert not self.bad, "\n".join(self.bad)


def _room(n, camera, views, scale_add=0.0, opacity_add=0.0, seed=43):
    from splat_slam_amd import synthetic as syn
    intr = syn.INTRINSICS[camera]
    params = syn.room_parameters(n, seed=seed, device=DEV)
    if scale_add:
        params["scaling"] = params["scaling"] + scale_add
    if opacity_add:
        params["opacity"] = params["opacity"] + opacity_add
    cams = syn.make_views(params, views, intr, DEV, seed=seed)
    return syn, intr, params, cams


def _activated_inputs(gm):
    """What render() hands to the rasterizer (gaussian_renderer/__init__.py:89-111), as fp32-exact doubles on the CPU."""
    with torch.no_grad():
        d = lambda t: t.detach().float().cpu().double()
        n = gm.get_xyz.shape[0]
        return dict(means3D=d(gm.get_xyz), means2D=torch.zeros(n, 3, dtype=torch.float64), opacities=d(gm.get_opacity),
                    shs=d(gm.get_features), scales=d(gm.get_scaling), rotations=d(gm.get_rotation),
                    theta=torch.zeros(3, dtype=torch.float64), rho=torch.zeros(3, dtype=torch.float64))


def _oracle_settings(cam, intr, dtype=torch.float64):
    c = lambda t: t.detach().float().cpu().to(dtype)
    f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
    return O.OracleSettings(intr["H"], intr["W"], f32(math.tan(cam.FoVx * 0.5)), f32(math.tan(cam.FoVy * 0.5)),
                            torch.zeros(3, dtype=dtype), 1.0, c(cam.world_view_transform), c(cam.full_proj_transform),
                            c(cam.projection_matrix), 0, c(cam.camera_center), False, False)


def _check_radii(soft, hip, ref, what):
    d = (hip.long() - ref.long()).abs()
    big = d > 1          # ceil(3 sqrt(lambda)) of a splat a few mm from the camera plane: radius ~1e4 px, fp32 resolves it to ~1e-3 relative
    big &= (hip > 0) & (ref > 0)     # (one side 0: the cull itself flipped -- in_front / empty tile rectangle at the image border)
    soft.check(int((d > 0).sum()) <= 3 and bool((d[big].double() <= 2e-3 * ref[big].double()).all()),
               f"{what}: radii differ at {int((d > 0).sum())} of {int((ref > 0).sum())} visible Gaussians (max {int(d.max())}: "
               f"{int(hip[d.argmax()])} vs {int(ref[d.argmax()])})")


def _check_image(soft, a, b, what):
    n_out, emax, m = outlier_report(a, b, REL)
    soft.check(n_out <= max(2, a.numel() // 2000), f"{what}: {n_out} pixels beyond {REL} (max err {emax:.3e}, max {m:.3e})")
    soft.check(emax <= 1.01 / 255.0 * max(m, 1.0), f"{what}: max err {emax:.3e}")


# ------------------------------------------------------------------------------------------------ (a) autograd API
def _strict_rel(a, b, knife_rows, width):
    """(rel. error over the Gaussians that are NOT on a knife edge, rel. error over those that are, how many of those exceed
    REL) -- all normalised by the largest reference value of the whole tensor."""
    a, b = a.double().reshape(-1, width), b.double().reshape(-1, width)
    m = b.abs().max().clamp_min(1e-30)
    e = (a - b).abs().max(dim=1).values / m
    strict = e[~knife_rows].max().item() if bool((~knife_rows).any()) else 0.0
    loose = e[knife_rows].max().item() if bool(knife_rows.any()) else 0.0
    return strict, loose, int((e[knife_rows] > REL).sum())


def _per_gaussian_report(soft, what, a, b, knife_rows, width, visible=None):
    """Per-Gaussian relative error (normalised by the tensor's largest reference value): quantiles over the visible Gaussians,
    and the two facts a tolerance on a norm cannot show: no Gaussian OFF the oracle's knife-edge list is beyond REL, and the
    Gaussians beyond REL are a subset of that list (a real defect could not hide behind it)."""
    a, b = a.double().reshape(-1, width), b.double().reshape(-1, width)
    m = b.abs().max().clamp_min(1e-30)
    e = (a - b).abs().max(dim=1).values / m
    rows = visible if visible is not None else (b.abs().max(dim=1).values > 0) | (a.abs().max(dim=1).values > 0)
    ev = e[rows]
    if ev.numel() == 0:
        return
    q = torch.quantile(ev, torch.tensor([0.5, 0.9, 0.99, 0.999], dtype=torch.float64)).tolist()
    beyond = e > REL
    off_list = beyond & ~knife_rows
    soft.check(int(off_list.sum()) == 0, f"{what}: per-Gaussian rel err quantiles 50/90/99/99.9 % = {q[0]:.1e}/{q[1]:.1e}/{q[2]:.1e}/{q[3]:.1e}, max "
                                         f"{ev.max().item():.1e}; {int(beyond.sum())} Gaussians beyond {REL} of which {int(off_list.sum())} are NOT on "
                                         f"the oracle's knife-edge list ({int(knife_rows.sum())} listed)")
    soft.check(q[0] <= 1e-5, f"{what}: median per-Gaussian rel err {q[0]:.1e}")


def _per_gaussian_bounded(soft, what, a, b, knife_rows, width, visible, max_frac=1e-3, hard=2e-3):
    """The acceptance rule of the long-list cases (round 4; it replaces "rel L2 <= 2e-4 and rel max <= 5e-3"): EVERY visible Gaussian
    within REL of the oracle, except a counted few -- at most max(3, max_frac x visible) -- that a flipped cut-off at one of their
    ~1000 pixels moved, and those by no more than `hard`.  The rule does not lean on the oracle's knife-edge list: on a
    surface-covering map that list holds practically every Gaussian (each has SOME pixel on a cut-off) and therefore pins
    nothing; its coverage is printed, and where it covers no more than a quarter of the visible Gaussians the exceptions must
    additionally all be on it."""
    a, b = a.double().reshape(-1, width), b.double().reshape(-1, width)
    m = b.abs().max().clamp_min(1e-30)
    e = (a - b).abs().max(dim=1).values / m
    nvis = int(visible.sum())
    ev = e[visible]
    if ev.numel() == 0:
        return
    q = torch.quantile(ev, torch.tensor([0.5, 0.99, 0.999], dtype=torch.float64)).tolist()
    beyond = e > REL
    allowed = max(3, int(math.ceil(max_frac * nvis)))
    cover = float((knife_rows & visible).sum()) / max(1, nvis)
    soft.check(int(beyond.sum()) <= allowed and e.max().item() <= hard,
               f"{what}: {int(beyond.sum())} of {nvis} visible Gaussians beyond {REL} (allowed {allowed}), worst {e.max().item():.2e} (allowed {hard}); "
               f"quantiles 50/99/99.9 % = {q[0]:.1e}/{q[1]:.1e}/{q[2]:.1e}; the oracle's knife-edge list covers {100 * cover:.1f} % of the visible Gaussians"
               + (" -- it pins nothing here, the count bound is the check" if cover > 0.25 else ""))
    soft.check(q[2] <= 0.5 * REL and q[0] <= 1e-5, f"{what}: 99.9 % quantile {q[2]:.1e}, median {q[0]:.1e}")
    if cover <= 0.25:
        soft.check(int((beyond & ~knife_rows).sum()) == 0, f"{what}: {int((beyond & ~knife_rows).sum())} Gaussians beyond {REL} are NOT on the knife-edge list")


WIDTH = {"means3D": 3, "means2D": 3, "opacities": 1, "shs": 3, "scales": 3, "rotations": 4}


@pytest.mark.parametrize("case", [("configs0", 20000, "replica", 0.0, True), ("configs1", 300000, "metric", 0.0, True),
                                  ("configs0_own_sort", 20000, "replica", 0.0, "own"), ("configs1_own_sort", 300000, "metric", 0.0, "own"),
                                  ("configs1_raw", 300000, "metric", 0.0, "raw"), ("configs1_opaque", 150000, "metric", 1.6, False)],
                         ids=lambda c: c[0])
def test_autograd_api_matches_oracle_at_config_size(case):
    name, n, camera, scale_add, deknife = case
    raw = deknife == "raw"          # the scene as generated: knife-edge Gaussians stay, and must be the ONLY ones beyond the tolerance
    own = deknife == "own"          # additionally de-tied in depth, the oracle sorts by its OWN depths (no hand-over of
    deknife = deknife is True or own    # the HIP forward's keys), nothing is excused, pose gradients at 1 x REL like everything else
    syn, intr, params, cams = _room(n, camera, 1, scale_add=scale_add)
    gm = syn.model_from_parameters(params, device=DEV)
    inp = _activated_inputs(gm)
    s = _oracle_settings(cams[0], intr)
    soft = Soft()
    if own:
        rounds, moves = move_off_knife_edges_and_depth_ties(inp, s)
        soft.check(True, f"{name}: {rounds} rounds, {moves} Gaussians pushed apart in depth: no knife edge, no near tie of depth left")
    elif deknife:      # the two headline configs are compared WITHOUT exceptions: first move the scene off its knife edges
        rounds = move_off_knife_edges(inp, s)
        soft.check(True, f"{name}: {rounds} rounds of nudging (opacity; centre and scale for tile-rectangle or radius edges) to clear the knife edges")
    g = torch.Generator().manual_seed(5)
    wc = torch.randn(3, intr["H"], intr["W"], generator=g, dtype=torch.float64)
    wd = torch.randn(1, intr["H"], intr["W"], generator=g, dtype=torch.float64)
    hip_out, hip_g, keys = run_hip(inp, s, wc, wd, want_depth_keys=True)
    view = s.viewmatrix.double().t()
    check_depth_keys(keys, hip_out[1], inp["means3D"] @ view[2, :3] + view[2, 3])
    knife = {}
    ref_out, ref_g = run_oracle(inp, s, wc, wd, dtype=torch.float64, depth_sort_key=None if own else keys, knife=knife)
    on_edge = knife_ids(knife, n)
    nvis = int((ref_out[1] > 0).sum())
    assert nvis > (1000 if n < 100000 else 10000), "scene is not visible enough to mean anything"
    if own:
        soft.check(torch.equal(hip_out[1], ref_out[1]), f"{name}: radii differ at {int((hip_out[1] != ref_out[1]).sum())} Gaussians (exact match required)")
    else:
        _check_radii(soft, hip_out[1], ref_out[1], name)
    if deknife:
        soft.check(int(on_edge.sum()) == 0, f"{name}: {int(on_edge.sum())} Gaussians still on a knife edge")
        for i, what in ((0, "color"), (2, "depth"), (3, "opacity")):
            r = rel_linf(hip_out[i], ref_out[i])
            soft.check(r <= REL, f"{name}/{what}: rel err {r:.3e} (no outlier pixels allowed)")
        soft.check(torch.equal(hip_out[4].long(), ref_out[4].long()), f"{name}: n_touched differs by {(hip_out[4].long() - ref_out[4].long()).abs().sum().item()} counts")
    else:
        # a splat of a surface-covering map spans ~1000 pixels: practically every Gaussian has SOME pixel on a knife edge, so
        # here the flips cannot be side-stepped; they are few per Gaussian, which the L2 error shows (and the max error bounds)
        soft.check(True, f"{name}: {int(on_edge.sum())} of {nvis} visible Gaussians have a pixel on a knife edge")
        for i, what in ((0, "color"), (2, "depth"), (3, "opacity")):
            _check_image(soft, hip_out[i], ref_out[i], f"{name}/{what}")
        nt, rnt = hip_out[4].long(), ref_out[4].long()
        soft.check((nt - rnt).abs().sum().item() <= max(2, nt.numel() // 500), f"{name}: n_touched differs by {(nt - rnt).abs().sum().item()} counts")
    vis_rows = ref_out[1] > 0
    for k in GRAD_KEYS:
        if raw and k in WIDTH:
            # every Gaussian the oracle does NOT list is held to REL; the listed ones are bounded and few of them may exceed REL
            strict, loose, n_loose = _strict_rel(hip_g[k], ref_g[k], on_edge, WIDTH[k])
            soft.check(strict <= REL, f"{name}: grad {k} rel err {strict:.3e} over the {int((~on_edge & vis_rows).sum())} visible Gaussians off the list")
            soft.check(loose <= 2e-2 and n_loose <= int(on_edge.sum()), f"{name}: grad {k}: {n_loose} of {int(on_edge.sum())} listed Gaussians beyond {REL}, worst {loose:.3e}")
            _per_gaussian_report(soft, f"{name}: grad {k}", hip_g[k], ref_g[k], on_edge, WIDTH[k], vis_rows)
        elif raw:
            r = rel_linf(hip_g[k].reshape(-1), ref_g[k].reshape(-1))
            soft.check(r <= 3 * REL, f"{name}: grad {k} rel err {r:.3e}")
        elif not deknife:
            if k in WIDTH:
                _per_gaussian_bounded(soft, f"{name}: grad {k}", hip_g[k], ref_g[k], on_edge, WIDTH[k], vis_rows)
            else:       # pose gradients are sums over all Gaussians: a flipped pair moves them by its share
                r = rel_linf(hip_g[k].reshape(-1), ref_g[k].reshape(-1))
                soft.check(r <= 3 * REL, f"{name}: grad {k} rel err {r:.3e}")
        elif k in WIDTH:
            strict, loose, n_loose = _strict_rel(hip_g[k], ref_g[k], on_edge, WIDTH[k])
            soft.check(strict <= REL, f"{name}: grad {k} rel err {strict:.3e} (Gaussians off the knife edges)")
            soft.check(loose <= 2e-2 and n_loose <= max(3, nvis // 200),
                       f"{name}: grad {k}: {n_loose} knife-edge Gaussians beyond {REL}, worst {loose:.3e}")
        else:       # pose gradients are sums over all Gaussians: a flipped pair moves them by its share
            r = rel_linf(hip_g[k].reshape(-1), ref_g[k].reshape(-1))
            soft.check(r <= (REL if deknife else 3 * REL), f"{name}: grad {k} rel err {r:.3e}")
    soft.done()


# ------------------------------------------------------------------------------------------------ (b) batched mapping path
def _decode_code_bytes(vb, H, W):
    """Signs the fused loss epilogue of blend_fwd left for blend_bwd<true>: [4, H, W] in {-1, 0, +1} (r, g, b, depth).
    One byte per pixel, 2 bits per value (1 = +, 2 = -), stored tile-major (8x8 tile * 64 + lane), sgr_blend.hip."""
    gx, gy = (W + 7) // 8, (H + 7) // 8
    raw = vb.d_color.view(-1).view(torch.uint8)[: gx * gy * 64].cpu().view(gy, gx, 8, 8).long()
    img = raw.permute(0, 2, 1, 3).reshape(gy * 8, gx * 8)[:H, :W]
    out = torch.zeros(4, H, W, dtype=torch.float64)
    for c in range(4):
        bits = (img >> (2 * c)) & 3
        out[c] = (bits == 1).double() - (bits == 2).double()
    return out


def _run_batched_case(n, camera, nviews, scale_add=0.0, opacity_add=0.0, min_long_tiles=0, min_huge_tiles=0, opacity_const=None,
                      by_l2=False, dump=None, prepare=None, own_depth_sort=False):
    """prepare(syn, intr, params, cams): edits the raw parameters before the loops are built.  own_depth_sort: the oracle sorts by its
    OWN depths (no hand-over of the HIP forward's keys) and nothing is excused as a knife edge."""
    import ctypes as C
    from splat_slam_amd import _native as nat
    from splat_slam_amd.fused import FusedMappingLoop
    syn, intr, params, cams = _room(n, camera, nviews, scale_add=scale_add, opacity_add=opacity_add)
    if prepare is not None:
        prepare(syn, intr, params, cams)
    if opacity_const is not None:       # every splat equally faint: nothing terminates early, whole lists are walked
        params["opacity"] = torch.full_like(params["opacity"], math.log(opacity_const / (1.0 - opacity_const)))
        cams = syn.make_views(params, nviews, intr, DEV, seed=43)
    H, W = intr["H"], intr["W"]
    lib = nat.lib()

    def run(fused_blend):
        """ONE sgr_map_views batch (forward + loss epilogue + backward + gather) with the tile kernels fused or not."""
        lib.sgr_set_option(nat.SGR_OPT_FUSED_BLEND, fused_blend)
        try:
            f = FusedMappingLoop(syn.DEFAULT_CONFIG, device=DEV)
            f.gaussians = syn.model_from_parameters(params, device=DEV)
            f.viewpoints = {c.uid: c for c in cams}
            f.current_window = list(range(nviews))
            f.build_keyframe_optimizers()
            for k, c in enumerate(cams):                         # exercise the exposure affine (slam_utils.py:72-75)
                c.exposure_a.data.fill_(0.04 * k - 0.03)
                c.exposure_b.data.fill_(0.01 - 0.008 * k)
            f._ensure_state()
            f._activate()
            f._run_views(cams, stats=True)
            torch.cuda.synchronize()
        finally:
            lib.sgr_set_option(nat.SGR_OPT_FUSED_BLEND, 1)
        return f

    # the un-fused pair leaves the loss-gradient signs in HBM (one code byte per pixel): read them, then run the path
    # bench.py times (the fused tile kernel) and require the SAME bits from it
    f0 = run(0)
    signs = [_decode_code_bytes(f0._views[c.uid], H, W) for c in cams]
    flat0 = f0._acc["flat"].clone()
    loss0 = torch.cat([f0._views[c.uid].loss for c in cams]).clone()
    del f0
    f = run(1)
    soft = Soft()
    soft.check(torch.equal(flat0, f._acc["flat"]), "fused tile kernel == blend_fwd -> code bytes -> blend_bwd<true>, bit for bit (gradients)")
    soft.check(torch.equal(loss0, torch.cat([f._views[c.uid].loss for c in cams])), "fused tile kernel == un-fused pair, bit for bit (losses)")
    gm, acc = f.gaussians, f._acc
    alpha = float(syn.DEFAULT_CONFIG["mapping"]["Training"]["alpha"])
    thr = float(syn.DEFAULT_CONFIG["mapping"]["Training"]["rgb_boundary_threshold"])

    # walked-list histogram of the last view: the long-list code paths must really have run
    vb = f._views[cams[-1].uid]
    ws = nat.SgrWorkspace(vb.saved.data_ptr(), vb.saved.numel(), vb.scratch.data_ptr(), vb.scratch.numel(), f._cap)
    hist = (C.c_int64 * 8)()
    nat.check(lib.sgr_query_list_histogram(C.byref(ws), n, H, W, hist, torch.cuda.current_stream().cuda_stream), "hist")
    long_tiles = int(hist[6]) + int(hist[7])
    soft.check(long_tiles >= min_long_tiles and int(hist[7]) >= min_huge_tiles,
               f"tiles by walked list length (0, 1-4, 5-8, 9-16, 17-32, 33-64, 65-256, >256): {list(hist)}")

    # ---- oracle: sum over the views of autograd through the reference's mapping loss
    inp = _activated_inputs(gm)
    x = {k: v.clone().requires_grad_(True) for k, v in inp.items() if k not in ("theta", "rho")}
    stat_accum = torch.zeros(n, dtype=torch.float64)
    stat_denom = torch.zeros(n, dtype=torch.float64)
    stat_maxr = torch.zeros(n, dtype=torch.float64)
    knife, seen = {}, torch.zeros(n, dtype=torch.bool)
    flip_px = []
    for k, cam in enumerate(cams):
        vb = f._views[cam.uid]
        s = _oracle_settings(cam, intr)
        x["means2D"].grad = None
        keys = hip_depth_keys(vb.saved, f._cap, n, H, W, vb.radii)
        view = s.viewmatrix.t()
        check_depth_keys(keys, vb.radii.cpu(), inp["means3D"] @ view[2, :3] + view[2, 3])
        col, radii, dep, opa, nt = O.rasterize(x["means3D"], x["means2D"], x["opacities"], shs=x["shs"], scales=x["scales"],
                                               rotations=x["rotations"], settings=s, depth_sort_key=None if own_depth_sort else keys,
                                               knife=None if own_depth_sort else knife)
        if own_depth_sort:
            soft.check(torch.equal(vb.radii.cpu().long(), radii.long()), f"view {k}: radii (exact)")
        _check_radii(soft, vb.radii.cpu(), radii, f"view {k}")
        a = torch.tensor(float(cam.exposure_a.item()), dtype=torch.float64, requires_grad=True)
        b = torch.tensor(float(cam.exposure_b.item()), dtype=torch.float64, requires_grad=True)
        gt = cam.original_image.detach().cpu().double()
        gtd = cam.depth.detach().cpu().double()[None]
        m = (gt.sum(dim=0, keepdim=True) > thr).double()
        md = (gtd > 0.01).double()
        r_rgb = (torch.exp(a) * col + b) * m - gt * m            # slam_utils.py:72-75,94-95
        r_dep = dep * md - gtd * md                              # :102-103
        loss = alpha * r_rgb.abs().mean() + (1 - alpha) * r_dep.abs().mean()
        soft.check(abs(vb.loss.item() - loss.item()) <= 2e-5 * abs(loss.item()), f"view {k}: loss {vb.loss.item():.8f} vs {loss.item():.8f}")
        # signs: HIP's code bytes must equal the oracle's wherever the residual is not a rounding knife edge -- except at
        # the handful of pixels where a cut-off (alpha >= 1/255, T >= 1e-4) fell on the other side in fp32 and moved the
        # pixel itself by up to 1/255 (the same allowance the image comparisons make)
        sg_hip = signs[k]
        r_all = torch.cat([r_rgb, r_dep]).detach()
        sg_ref = torch.sign(r_all)
        firm = r_all.abs() >= KNIFE
        n_flip = int((sg_hip[firm] != sg_ref[firm]).sum())
        flip_px.append(torch.nonzero(firm & (sg_hip != sg_ref)).tolist())
        soft.check(n_flip <= 4 + r_all.numel() // 200000, f"view {k}: {n_flip} loss-gradient signs differ away from the knife edge "
                   f"(largest |residual| among them {float(r_all[firm & (sg_hip != sg_ref)].abs().max()) if n_flip else 0.0:.2e})")
        soft.check(int((~firm & (r_all != 0)).sum()) < max(100, 0.002 * r_all.numel()), f"view {k}: {int((~firm & (r_all != 0)).sum())} knife-edge residuals")
        sg = torch.where(firm & (sg_hip == sg_ref), sg_ref, sg_hip)
        surrogate = (alpha * (sg[:3] * r_rgb).sum() / (3 * H * W) + (1 - alpha) * (sg[3:] * r_dep).sum() / (H * W))
        surrogate.backward()                                      # gradient of the L1 loss with those signs; accumulates over views
        g2 = x["means2D"].grad[:, :2]
        vis = radii > 0
        seen |= vis
        stat_accum += torch.where(vis, g2.norm(dim=1), torch.zeros(n, dtype=torch.float64))
        stat_denom += vis.double()
        stat_maxr = torch.maximum(stat_maxr, radii.double())
        row = f._exp.row_of(cam)
        d_exp = f._exp.grad[row].cpu().double() if row is not None else vb.d_exp.cpu().double()
        ref_exp = torch.stack([a.grad, b.grad])
        soft.check(bool((d_exp - ref_exp).abs().max() <= 3e-4 * ref_exp.abs().max().clamp_min(1e-12)),
                   f"view {k}: exposure grad {d_exp.tolist()} vs {ref_exp.tolist()}")
        nt_h = vb.n_touched.cpu().long()
        soft.check((nt_h - nt.long()).abs().sum().item() <= max(2, n // 500), f"view {k}: n_touched differs by {(nt_h - nt.long()).abs().sum().item()}")

    on_edge = torch.zeros(n, dtype=torch.bool) if own_depth_sort else knife_ids(knife, n)      # (own_depth_sort: nobody is excused)
    nvis = int(seen.sum())
    soft.check(True, f"{int(on_edge.sum())} of {nvis} visible Gaussians sit on a knife edge in some view")
    pairs = (("xyz", "means3D", 3), ("f_dc", "shs", 3), ("opacity", "opacities", 1), ("scaling", "scales", 3), ("rotation", "rotations", 4))
    dbg = {}
    for mine, ref, w in pairs:
        a, b = acc[mine].detach().cpu().double().reshape(-1, w), x[ref].grad.double().reshape(-1, w)
        if by_l2:      # long lists: (almost) every Gaussian has some pixel on a knife edge -- see the opaque autograd case
            _per_gaussian_bounded(soft, f"accumulated grad {mine}", a, b, on_edge, w, seen)
            continue
        strict, loose, n_loose = _strict_rel(a, b, on_edge, w)
        e = (a - b).abs().max(dim=1).values / b.abs().max()
        e[on_edge] = 0
        worst = int(e.argmax())
        soft.check(strict <= REL, f"accumulated grad {mine}: rel err {strict:.3e} (Gaussians off the knife edges; worst: Gaussian {worst}, "
                                  f"hip {a[worst].tolist()} vs oracle {b[worst].tolist()})")
        if strict > REL and mine == "xyz":          # what does that Gaussian look like in every view?
            for k, cam in enumerate(cams):
                pp = O.preprocess(inp["means3D"], None, inp["opacities"], inp["shs"], None, inp["scales"], inp["rotations"], None, None,
                                  None, _oracle_settings(cam, intr))
                soft.check(True, f"  view {k}: Gaussian {worst} radius {int(pp.radii[worst])} xy {pp.xy[worst].tolist()} depth {float(pp.depth[worst]):.5f} "
                                 f"conic {pp.conic[worst].tolist()} opacity {float(pp.opacity[worst]):.6f} rect {pp.rect[worst].tolist()} hip radius "
                                 f"{int(f._views[cam.uid].radii[worst])}; sign-flip pixels (c, y, x) of this view: {flip_px[k]}")
            soft.check(True, f"  scales {inp['scales'][worst].tolist()} rot {inp['rotations'][worst].tolist()} xyz {inp['means3D'][worst].tolist()}")
        soft.check(loose <= 2e-2 and n_loose <= max(3, nvis // 100), f"accumulated grad {mine}: {n_loose} knife-edge Gaussians beyond {REL}, worst {loose:.3e}")
        top = torch.topk(e, 20).indices
        dbg[mine + "_idx"], dbg[mine + "_hip"], dbg[mine + "_ref"] = top.numpy(), a[top].numpy(), b[top].numpy()
    if dump and dbg:
        import numpy as np
        import os
        os.makedirs(os.path.dirname(dump), exist_ok=True)
        np.savez_compressed(dump, on_edge=torch.nonzero(on_edge).flatten().numpy(), **dbg,
                            **{f"cam{k}_view": _oracle_settings(c, intr).viewmatrix.numpy() for k, c in enumerate(cams)})
    a, b = gm.xyz_gradient_accum.cpu().double().reshape(-1, 1), stat_accum.reshape(-1, 1)
    if by_l2:
        _per_gaussian_bounded(soft, "densification statistic", a, b, on_edge, 1, seen)
    else:
        strict, loose, n_loose = _strict_rel(a, b, on_edge, 1)
        soft.check(strict <= REL and loose <= 2e-2, f"densification statistic: rel err {strict:.3e} / knife-edge {loose:.3e}")
    soft.check(int((gm.denom.cpu().reshape(-1).double() != stat_denom).sum()) <= 3, "denom differs")
    bad_r = (gm.max_radii2D.cpu().double() - stat_maxr).abs() > torch.clamp_min(2e-3 * stat_maxr, 1.0)
    soft.check(int(bad_r.sum()) <= 3, f"max_radii2D differs at {int(bad_r.sum())} Gaussians")
    soft.done()
    return list(hist)


def test_batched_mapping_path_matches_oracle_configs1():
    _run_batched_case(300000, "metric", 4, dump="gpurun_out/fullsize_configs1_dbg.npz")


def test_batched_mapping_path_matches_oracle_configs0():
    _run_batched_case(20000, "replica", 3)


def test_batched_mapping_path_matches_oracle_opaque_scene():
    # a converged, surface-covering map: lists beyond 64 entries (LDS sort, multi-chunk backward) at 640x480
    _run_batched_case(150000, "metric", 2, scale_add=1.6, min_long_tiles=50, by_l2=True)


def test_batched_mapping_path_matches_oracle_very_long_lists():
    # the whole 300 k map seen through a 96x64 camera with enlarged, uniformly faint splats (opacity 0.03): hundreds of pairs
    # per 8x8 tile and nothing terminates early, so the walked lists run past 256 entries (1024- / 4096-key LDS sort builds,
    # multi-chunk backward with carries)
    _run_batched_case(300000, "tiny", 4, scale_add=3.2, opacity_const=0.02, min_long_tiles=40, min_huge_tiles=10, by_l2=True)
