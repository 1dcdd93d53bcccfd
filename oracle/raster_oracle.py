"""CPU oracle for the Splat-SLAM rasterizer boundary  --  TEST INFRASTRUCTURE ONLY.

This file is the parity checker for the HIP path. Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``diff_gaussian_rasterization`` / ``splat_slam_amd``) never does.

PARITY UNPINNED: the arithmetic of this path lives in the un-vendored submodule
``rmurai0610/diff-gaussian-rasterization-w-pose`` @ 43e21bff91cd24986ee3dd52fe0bb06952e50ec7
(+ near-plane patch 0.2 -> 0.001; /root/reference/README.md:88-92, Dockerfile:20).
Its sources are absent from /root/reference (empty dir, .gitmodules:4-6) and the
reference ships no tests or golden vectors for it, so this restatement follows the
published 3DGS / MonoGS algorithm and is pinned only by
  * the reference's own call site and conventions
      thirdparty/gaussian_splatting/gaussian_renderer/__init__.py:58-74,130-141  (arguments, 5-tuple result)
      thirdparty/monogs/utils/camera_utils.py:94-108                            (transposed matrices)
      thirdparty/gaussian_splatting/utils/graphics_utils.py:72-93               (projection, w_clip = z_view)
      thirdparty/gaussian_splatting/utils/sh_utils.py:24-118                    (SH basis, +0.5 offset)
      thirdparty/gaussian_splatting/utils/general_utils.py:113-136              (quaternion (w,x,y,z) -> R)
      thirdparty/monogs/utils/pose_utils.py:66-98                               (tau = [rho, theta], left perturbation)
  * closed-form known answers and fp64 finite differences (tests/test_oracle_*.py).

Everything is written with differentiable torch ops: gradients (including the camera
pose gradient wrt ``theta``/``rho``) come from autograd.  Upstream's hand-written
backward deviates from the exact derivative in three documented places; each is a named
switch so it can be flipped if the real CUDA build is ever available:

  CLAMP_STRAIGHT_THROUGH  alpha = min(0.99, o*G) is treated as identity in the backward
  FRUSTUM_CLAMP_CONST     the +-1.3*tanfov clamp of t.x/t.y yields a constant (no d/dt.z)
  CAMPOS_CONST            camera centre used for SH view directions carries no pose grad

A fourth switch is OFF by default: UPSTREAM_POSE_JACOBIAN.  Upstream's 2D-mean -> pose term is believed to
read five scalars of projmatrix_raw (P00, P11, P22, P23, P32) and to use d x_ndc / d p_cam = (P00/w, 0, -x_hom/w^2), which
drops the principal-point terms P02/w, P12/w: exact for cx = W/2, cy = H/2, off by O(|P02|) otherwise (8e-4 on Replica;
irrelevant for default runs, mapping.BA False).  Default = the exact derivative; the HIP library has the same switch
(SGR_OPT_UPSTREAM_POSE_JACOBIAN) and tests/ check both settings on both sides.
"""
from __future__ import annotations

import math
from typing import NamedTuple

import torch

NEAR_PLANE = 0.001          # README.md:88-92 (patched from 0.2)
TILE = 16                   # upstream BLOCK_X = BLOCK_Y
ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.99
T_EPS = 1e-4
DILATION = 0.3
N_TOUCHED_T = 0.5           # n_touched counts pixels composited while T' > 0.5
CLAMP_STRAIGHT_THROUGH = True
FRUSTUM_CLAMP_CONST = True
CAMPOS_CONST = True
UPSTREAM_POSE_JACOBIAN = False   # True: pose path of the projected mean without the principal-point terms (see the docstring)

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


class OracleSettings(NamedTuple):
    """Same 13 fields, same order, as the settings built at gaussian_renderer/__init__.py:58-72."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    projmatrix_raw: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _hat(v):
    z = torch.zeros((), dtype=v.dtype, device=v.device)
    return torch.stack([torch.stack([z, -v[2], v[1]]),
                        torch.stack([v[2], z, -v[0]]),
                        torch.stack([-v[1], v[0], z])])


def se3_exp_small(rho, theta):
    """SE3_exp of pose_utils.py:66-78 in its ``angle < 1e-5`` branch (:38-45, :55-62)."""
    W = _hat(theta)
    W2 = W @ W
    I = torch.eye(3, dtype=rho.dtype, device=rho.device)
    R = I + W + 0.5 * W2
    V = I + 0.5 * W + W2 / 6.0
    T = torch.eye(4, dtype=rho.dtype, device=rho.device)
    T = T.clone()
    T[:3, :3] = R
    T[:3, 3] = V @ rho
    return T


def quat_to_rot(q):
    """general_utils.py:113-136 without the normalisation (upstream kernel uses q as given)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.view(-1, 3, 3)


def eval_sh_rgb(deg, sh, dirs):
    """sh: [N, K, 3]; dirs: [N, 3] unit.  Basis of sh_utils.py:55-118 (deg <= 3)."""
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9]
                       + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13]
                       + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


class Preprocessed(NamedTuple):
    visible: torch.Tensor      # bool [N]  (radii > 0)
    radii: torch.Tensor        # int32 [N]
    xy: torch.Tensor           # [N,2] pixel-space centre
    depth: torch.Tensor        # [N]   view-space z
    conic: torch.Tensor        # [N,3] (A, B, C)
    opacity: torch.Tensor      # [N]
    rgb: torch.Tensor          # [N,3]
    rect: torch.Tensor         # int64 [N,4] (x0, y0, x1, y1) in 16x16 tiles, half-open


def preprocess(means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
               cov3D_precomp, theta, rho, s: OracleSettings) -> Preprocessed:
    dt = means3D.dtype
    dev = means3D.device
    N = means3D.shape[0]
    H, W = int(s.image_height), int(s.image_width)
    view = s.viewmatrix.to(dt).t()        # W2C (camera_utils.py:94-96 stores the transpose)
    proj_given = s.projmatrix.to(dt).t()  # P @ W2C (camera_utils.py:98-104)
    P = s.projmatrix_raw.to(dt).t()

    # pose perturbation W2C(tau) = SE3_exp(tau) @ W2C evaluated at tau = 0 (pose_utils.py:81-90:
    # the deltas are applied to R,T by update_pose, never to the matrix handed to the rasterizer)
    view_eff, proj_eff = view, proj_given
    if (theta is not None and theta.numel() == 3 and theta.requires_grad) or \
       (rho is not None and rho.numel() == 3 and rho.requires_grad):
        th = theta.to(dt).reshape(3) if theta is not None and theta.numel() == 3 else torch.zeros(3, dtype=dt)
        rh = rho.to(dt).reshape(3) if rho is not None and rho.numel() == 3 else torch.zeros(3, dtype=dt)
        E = se3_exp_small(rh - rh.detach(), th - th.detach())
        view_eff = E @ view
        P_pose = P
        if UPSTREAM_POSE_JACOBIAN:            # the pose path does not see the principal-point column entries
            P_pose = P.clone()
            P_pose[0, 2] = 0.0
            P_pose[1, 2] = 0.0
        pv = P_pose @ view_eff
        proj_eff = proj_given + (pv - pv.detach())

    p = means3D
    p_view = p @ view_eff[:3, :3].t() + view_eff[:3, 3]
    p_hom = p @ proj_eff[:, :3].t() + proj_eff[:, 3]
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    ndc = p_hom[:, :2] * p_w[:, None]
    if means2D is not None:
        ndc = ndc + (means2D[:, :2] - means2D[:, :2].detach())   # dummy leaf: receives dL/d(ndc)

    in_front = p_view[:, 2] > NEAR_PLANE

    # --- 3D covariance
    if cov3D_precomp is not None and cov3D_precomp.numel() > 0:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4],
                             c[:, 2], c[:, 4], c[:, 5]], dim=-1).view(N, 3, 3)
    else:
        R = quat_to_rot(rotations)
        M = R * (s.scale_modifier * scales)[:, None, :]       # R @ diag(s)
        Sigma = M @ M.transpose(1, 2)                         # gaussian_model.py:68-74

    # --- EWA projection
    tz = p_view[:, 2]
    tz_safe = torch.where(in_front, tz, torch.ones_like(tz))
    limx, limy = 1.3 * s.tanfovx, 1.3 * s.tanfovy
    txtz = p_view[:, 0] / tz_safe
    tytz = p_view[:, 1] / tz_safe
    if FRUSTUM_CLAMP_CONST:
        tx = torch.where((txtz < -limx) | (txtz > limx), (txtz.clamp(-limx, limx) * tz_safe).detach(), p_view[:, 0])
        ty = torch.where((tytz < -limy) | (tytz > limy), (tytz.clamp(-limy, limy) * tz_safe).detach(), p_view[:, 1])
    else:
        tx = txtz.clamp(-limx, limx) * tz_safe
        ty = tytz.clamp(-limy, limy) * tz_safe
    fx = W / (2.0 * s.tanfovx)
    fy = H / (2.0 * s.tanfovy)
    zeros = torch.zeros_like(tz_safe)
    J = torch.stack([fx / tz_safe, zeros, -fx * tx / (tz_safe * tz_safe),
                     zeros, fy / tz_safe, -fy * ty / (tz_safe * tz_safe)], dim=-1).view(N, 2, 3)
    Tm = J @ view_eff[:3, :3]                                  # [N,2,3]
    cov2 = Tm @ Sigma @ Tm.transpose(1, 2)
    a = cov2[:, 0, 0] + DILATION
    b = cov2[:, 0, 1]
    c_ = cov2[:, 1, 1] + DILATION
    det = a * c_ - b * b
    det_ok = det != 0
    det_safe = torch.where(det_ok, det, torch.ones_like(det))
    conic = torch.stack([c_ / det_safe, -b / det_safe, a / det_safe], dim=-1)
    mid = 0.5 * (a + c_)
    disc = torch.clamp_min(mid * mid - det, 0.1).sqrt()
    lam = torch.maximum(mid + disc, mid - disc)
    radius = torch.ceil(3.0 * lam.detach().clamp_min(0).sqrt())

    xy = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=-1)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    xyd = xy.detach()

    def _trunc_clamp(v, hi):
        return torch.clamp(torch.trunc(v), 0, hi).to(torch.int64)

    x0 = _trunc_clamp((xyd[:, 0] - radius) / TILE, gx)
    y0 = _trunc_clamp((xyd[:, 1] - radius) / TILE, gy)
    x1 = _trunc_clamp((xyd[:, 0] + radius + TILE - 1) / TILE, gx)
    y1 = _trunc_clamp((xyd[:, 1] + radius + TILE - 1) / TILE, gy)
    area = (x1 - x0) * (y1 - y0)
    finite = torch.isfinite(xyd).all(dim=1) & torch.isfinite(radius)
    visible = in_front & det_ok & (area > 0) & finite
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)

    # --- colour
    if colors_precomp is not None and colors_precomp.numel() > 0:
        rgb = colors_precomp
    else:
        campos = s.campos.to(dt)
        if not CAMPOS_CONST:
            raise NotImplementedError
        d = p - campos.detach()
        d = d / d.norm(dim=1, keepdim=True)
        rgb = eval_sh_rgb(int(s.sh_degree), shs, d) + 0.5
        rgb = torch.clamp_min(rgb, 0.0)

    rect = torch.stack([x0, y0, x1, y1], dim=-1)
    return Preprocessed(visible, radii, xy, tz, conic, opacities.reshape(N), rgb, rect)


def build_tile_lists(pp: Preprocessed, H: int, W: int, depth_sort_key=None):
    """(tile, depth, idx)-ordered duplication list: returns (tile_ids[R], gauss_ids[R]) sorted.

    depth_sort_key (optional, fp32 [N]): the sort key of upstream is the fp32 BIT PATTERN of the view-space depth as the
    kernel computed it.  Two Gaussians whose depths agree to the last ulp or two are therefore ordered by fp32 rounding;
    a comparison against another implementation can hand that implementation's fp32 depths in here so that such near
    ties are broken the same way (the caller checks that the keys equal this oracle's depths to a few ulp)."""
    gx = (W + TILE - 1) // TILE
    vis = torch.nonzero(pp.visible).flatten()
    if vis.numel() == 0:
        e = torch.zeros(0, dtype=torch.int64)
        return e, e
    r = pp.rect[vis]
    w = r[:, 2] - r[:, 0]
    h = r[:, 3] - r[:, 1]
    cnt = w * h
    owner = torch.repeat_interleave(torch.arange(vis.numel()), cnt)
    start = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(int(cnt.sum())) - start[owner]
    ty = r[owner, 1] + local // w[owner]
    tx = r[owner, 0] + local % w[owner]
    tile = ty * gx + tx
    gid = vis[owner]
    # stable sort by depth (fp32 bit pattern, like the upstream key), then stable by tile
    dkey = (pp.depth.detach().to(torch.float32) if depth_sort_key is None else depth_sort_key.to(torch.float32))[gid]
    o1 = torch.argsort(dkey, stable=True)
    tile, gid = tile[o1], gid[o1]
    o2 = torch.argsort(tile, stable=True)
    return tile[o2], gid[o2]


KNIFE_BAND = 5e-4          # relative half-width of "fp32 rounding may decide this cut-off" for alpha (see knife_edge_gaussians)
KNIFE_PIXEL_ERR = 4e-4     # ... or, if larger, what this much error (in pixels) of the projected centre does to alpha
KNIFE_CONIC_EPS = 1e-6     # ... or what ~8 fp32 ulps of the quadratic form's TERMS do to the exponent
KNIFE_BAND_T = 1e-4        # ... and for the transmittance thresholds (products of a few (1 - alpha): far better conditioned)


def blend(pp: Preprocessed, tile_ids, gauss_ids, s: OracleSettings, dt, knife=None):
    H, W = int(s.image_height), int(s.image_width)
    gx = (W + TILE - 1) // TILE
    bg = s.bg.to(dt).reshape(3)
    N = pp.xy.shape[0]
    color = bg.reshape(3, 1, 1).expand(3, H, W).clone()
    depth = torch.zeros(1, H, W, dtype=dt)
    opac = torch.zeros(1, H, W, dtype=dt)
    n_touched = torch.zeros(N, dtype=torch.int64)
    if tile_ids.numel() == 0:
        return color, depth, opac, n_touched.to(torch.int32)
    uniq, counts = torch.unique_consecutive(tile_ids, return_counts=True)
    starts = torch.cumsum(counts, 0) - counts
    col_parts, dep_parts, opa_parts, idx_parts = [], [], [], []
    for t, st, cn in zip(uniq.tolist(), starts.tolist(), counts.tolist()):
        ty, tx = divmod(t, gx)
        ids = gauss_ids[st:st + cn]
        ys = torch.arange(ty * TILE, min((ty + 1) * TILE, H))
        xs = torch.arange(tx * TILE, min((tx + 1) * TILE, W))
        py, px = torch.meshgrid(ys, xs, indexing="ij")
        px = px.reshape(-1)
        py = py.reshape(-1)
        dx = pp.xy[ids, 0][None, :] - px[:, None].to(dt)
        dy = pp.xy[ids, 1][None, :] - py[:, None].to(dt)
        con = pp.conic[ids]
        power = -0.5 * (con[:, 0] * dx * dx + con[:, 2] * dy * dy) - con[:, 1] * dx * dy
        G = torch.exp(torch.clamp_max(power, 0.0))
        raw = pp.opacity[ids][None, :] * G
        if CLAMP_STRAIGHT_THROUGH:
            alpha = raw + (raw.clamp_max(ALPHA_MAX) - raw).detach()
        else:
            alpha = raw.clamp_max(ALPHA_MAX)
        keep = (power <= 0) & (alpha.detach() >= ALPHA_MIN)
        alpha = torch.where(keep, alpha, torch.zeros_like(alpha))
        one_m = 1.0 - alpha
        T_after = torch.cumprod(one_m, dim=1)
        T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], dim=1)
        live = keep & (T_after.detach() >= T_EPS)        # monotone: once below, stays below
        wgt = torch.where(live, alpha * T_before, torch.zeros_like(alpha))
        C = wgt @ pp.rgb[ids]
        D = wgt @ pp.depth[ids]
        # final transmittance = T after the last live entry
        T_live = torch.where(live, one_m, torch.ones_like(one_m)).prod(dim=1)
        col_parts.append(C + T_live[:, None] * bg[None, :])
        dep_parts.append(D)
        opa_parts.append(1.0 - T_live)
        idx_parts.append(py * W + px)
        touched = (live & (T_after.detach() > N_TOUCHED_T)).sum(dim=0)
        n_touched.index_add_(0, ids, touched)
        if knife is not None:
            # (pixel, splat) pairs that sit within KNIFE_BAND of a cut-off: alpha vs 1/255 (pixel centres carry ~1e-4 px of
            # fp32 error, which the exponent turns into up to ~1e-4 relative in alpha), T' vs 1e-4, T' vs 0.5 (n_touched)
            rawd, Td = raw.detach(), T_after.detach()
            # d(ln alpha) = |grad power| * d(centre): steep sub-pixel splats (conic ~ 1/0.3) far from the image origin turn
            # a few fp32 ulps of the projected centre (~1e-4 px at coordinate 500) into > 5e-4 relative in alpha
            dpx = (con[:, 0] * dx + con[:, 1] * dy).detach()
            dpy = (con[:, 1] * dx + con[:, 2] * dy).detach()
            band = torch.clamp_min(torch.sqrt(dpx * dpx + dpy * dpy) * KNIFE_PIXEL_ERR, KNIFE_BAND)
            # ... and the quadratic form of a long thin splat cancels: its three terms can be 1e4 while their sum is ~0, so fp32
            # (any fp32 implementation, upstream's included) carries an absolute error of ~8 ulps of the TERMS in `power`
            pw_err = ((con[:, 0] * dx * dx + con[:, 2] * dy * dy).abs() + 2.0 * (con[:, 1] * dx * dy).abs()).detach() * KNIFE_CONIC_EPS
            band = torch.maximum(band, pw_err)
            near = (power.detach() <= 0) & ((rawd * 255.0 - 1.0).abs() < band)
            # the `power > 0 -> skip` rule itself: exact power is never positive, a rounded one within pw_err of 0 may be
            near |= (power.detach() > -pw_err) & (rawd * 255.0 >= 1.0 - band)
            near |= keep & (T_before.detach() >= T_EPS) & (((Td / T_EPS - 1.0).abs() < KNIFE_BAND_T) | ((Td / N_TOUCHED_T - 1.0).abs() < KNIFE_BAND_T))
            if bool(near.any()):
                # ... and every splat composited at a pixel that has such a pair: a flip there changes their T / "colour
                # behind" at that pixel by up to 1/255 (measured: 5e-4 of the largest gradient for a small splat)
                hit = (keep & near.any(dim=1, keepdim=True)).any(dim=0) | near.any(dim=0)
                knife.setdefault("gaussians", []).append(ids[hit])
    idx = torch.cat(idx_parts)
    color = color.reshape(3, -1).index_copy(1, idx, torch.cat(col_parts).t()).reshape(3, H, W)
    depth = depth.reshape(-1).index_copy(0, idx, torch.cat(dep_parts)).reshape(1, H, W)
    opac = opac.reshape(-1).index_copy(0, idx, torch.cat(opa_parts)).reshape(1, H, W)
    return color, depth, opac, n_touched.to(torch.int32)


def rasterize(means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
              rotations=None, cov3D_precomp=None, theta=None, rho=None, *, settings: OracleSettings, depth_sort_key=None,
              knife=None):
    """Oracle for GaussianRasterizer.forward (call site gaussian_renderer/__init__.py:130-141).

    Returns (color[3,H,W], radii int32[N], depth[1,H,W], opacity[1,H,W], n_touched int32[N]).
    """
    dt = means3D.dtype
    pp = preprocess(means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
                    cov3D_precomp, theta, rho, settings)
    tile_ids, gauss_ids = build_tile_lists(pp, int(settings.image_height), int(settings.image_width), depth_sort_key)
    color, depth, opac, n_touched = blend(pp, tile_ids, gauss_ids, settings, dt, knife=knife)   # knife: see knife_edge_gaussians
    if knife is not None:
        knife.setdefault("gaussians", []).append(_geometric_knife_edges(pp, int(settings.image_height), int(settings.image_width)))
    return color, pp.radii, depth, opac, n_touched


@torch.no_grad()
def knife_edge_gaussians(means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, *,
                         settings: OracleSettings, detail=False):
    """Indices of the Gaussians that have, at some pixel, a composite / skip decision within KNIFE_BAND of its threshold.

    The rasterizer is piecewise continuous; exactly AT such a threshold two correct fp32 implementations (or fp32 and fp64)
    may decide differently, which moves that Gaussian's gradients by the whole contribution of the pixel (measured at
    640x320 / 20 k Gaussians: 2 such pairs per view, 3e-3 of the largest gradient).  Parity tests use this list to move the
    scene off the knife edge (nudging those opacities by a fraction of a per cent) so that every comparison can then be held
    to the tolerance without exceptions."""
    pp = preprocess(means3D, None, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, None, None, settings)
    tile_ids, gauss_ids = build_tile_lists(pp, int(settings.image_height), int(settings.image_width))
    knife = {}
    blend(pp, tile_ids, gauss_ids, settings, means3D.dtype, knife=knife)
    found = knife.get("gaussians", [])
    if detail:        # what a test needs to move the scene off the edges: opacity decides the first kind, geometry the second
        geo = _geometric_knife_edges(pp, int(settings.image_height), int(settings.image_width), sources_only=True)
        return {"alpha": torch.unique(torch.cat(found)) if found else torch.zeros(0, dtype=torch.int64), "geometric": geo}
    found.append(_geometric_knife_edges(pp, int(settings.image_height), int(settings.image_width)))
    return torch.unique(torch.cat(found))


KNIFE_RADIUS_REL = 2e-4    # 3 sqrt(lambda_max) this close (relative) to an integer: ceil() may round the other way in fp32


def _geometric_knife_edges(pp, H, W, sources_only=False):
    """The two integer decisions of preprocessCUDA (forward.cu:240-260 as restated in preprocess above) that fp32 rounding of the
    projected centre can flip: the 16-pixel tile rectangle trunc((xy -+ r [+ 15]) / 16) when xy -+ r is within KNIFE_PIXEL_ERR
    of a multiple of 16 (measured: centre x = 507.99997 in fp64, 508.0000x in fp32 -> one more column of tiles, whose pixels
    at dx = 4 still carry alpha = 0.011 -> 12 % of that Gaussian's gradient), and the radius ceil(3 sqrt(lambda_max))."""
    vis = pp.visible
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    r = pp.radii.to(pp.xy.dtype)
    hit = torch.zeros_like(vis)
    for axis, g in ((0, gx), (1, gy)):
        for u in ((pp.xy[:, axis] - r) / TILE, (pp.xy[:, axis] + r + TILE - 1) / TILE):
            u = u.detach()
            hit |= ((u - torch.round(u)).abs() < KNIFE_PIXEL_ERR / TILE) & (u > -0.5) & (u < g + 0.5)
    con = pp.conic.detach()
    detc = con[:, 0] * con[:, 2] - con[:, 1] * con[:, 1]
    detc = torch.where(detc != 0, detc, torch.ones_like(detc))
    a, c_ = con[:, 2] / detc, con[:, 0] / detc
    mid = 0.5 * (a + c_)
    lam = mid + torch.clamp_min(mid * mid - 1.0 / detc, 0.1).sqrt()
    ext = 3.0 * lam.clamp_min(0).sqrt()
    hit |= (ext - torch.round(ext)).abs() < KNIFE_RADIUS_REL * ext
    hit &= vis
    if sources_only:
        return torch.nonzero(hit).flatten()
    # ... and whoever shares pixels with such a splat: its flip puts alpha up to ~0.01 in front of (or takes it away from)
    # them at the pixels of the gained / lost tile row or column (measured: the neighbour 4 px away moved by 3e-4)
    out = hit.clone()
    vi = torch.nonzero(vis).flatten()
    vx, vy, vr = pp.xy[vi, 0].detach(), pp.xy[vi, 1].detach(), r[vi]
    for h in torch.nonzero(hit).flatten().tolist():
        reach = vr + r[h] + 1.0
        out[vi[((vx - pp.xy[h, 0].detach()).abs() <= reach) & ((vy - pp.xy[h, 1].detach()).abs() <= reach)]] = True
    return torch.nonzero(out).flatten()


# ----------------------------------------------------------------------------------------------
# camera helpers restating camera_utils.py:94-108 / graphics_utils.py:33-46,72-93 (CPU, any dtype)
# ----------------------------------------------------------------------------------------------
def projection_matrix(znear, zfar, cx, cy, fx, fy, W, H, dtype=torch.float32):
    """graphics_utils.py:72-93 (getProjectionMatrix2)."""
    left = ((2 * cx - W) / W - 1.0) * W / 2.0
    right = ((2 * cx - W) / W + 1.0) * W / 2.0
    top = ((2 * cy - H) / H + 1.0) * H / 2.0
    bottom = ((2 * cy - H) / H - 1.0) * H / 2.0
    left, right = znear / fx * left, znear / fx * right
    top, bottom = znear / fy * top, znear / fy * bottom
    P = torch.zeros(4, 4, dtype=dtype)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_settings(w2c, fx, fy, cx, cy, W, H, bg=None, sh_degree=0, znear=0.01, zfar=100.0,
                  scale_modifier=1.0, dtype=torch.float32) -> OracleSettings:
    """Settings exactly as render() builds them (gaussian_renderer/__init__.py:55-72) from a W2C pose."""
    w2c = w2c.to(dtype)
    P = projection_matrix(znear, zfar, cx, cy, fx, fy, W, H, dtype).t()     # mapper.py:841-850 (.transpose(0,1))
    view = w2c.t().contiguous()
    full = view @ P
    campos = torch.linalg.inv(view)[3, :3]
    fovx = 2 * math.atan(W / (2 * fx))
    fovy = 2 * math.atan(H / (2 * fy))
    if bg is None:
        bg = torch.zeros(3, dtype=dtype)
    return OracleSettings(H, W, math.tan(fovx * 0.5), math.tan(fovy * 0.5), bg.to(dtype), scale_modifier,
                          view, full, P.contiguous(), sh_degree, campos, False, False)
