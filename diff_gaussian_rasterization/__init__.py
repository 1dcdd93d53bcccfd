"""Drop-in for the `diff_gaussian_rasterization` package that Splat-SLAM imports at
/root/reference/thirdparty/gaussian_splatting/gaussian_renderer/__init__.py:15-18 -- same module name, same two
public symbols, same call signature (:58-74, :130-141), same 5-tuple result (:130) -- backed by the hand-written
gfx950 kernels in libsplat_hip.so through the C ABI of include/splat_hip.h.

The upstream package (rmurai0610/diff-gaussian-rasterization-w-pose @ 43e21bf) is CUDA; this one is MI355X only.
There is no CPU / eager fallback: CPU tensors or a missing library raise.
"""
from typing import NamedTuple

import ctypes as C
import os

import torch
import torch.nn as nn

from splat_slam_amd import _native as nat

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]


class GaussianRasterizationSettings(NamedTuple):
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


# ----------------------------------------------------------------------------------------------------------------
# workspace policy.  The library never allocates: torch's caching allocator owns everything.
#   saved   : one block per forward call (up to 12 forwards are outstanding before one backward, mapper.py:426-490)
#   scratch : one growing block per device, shared by all calls on the stream
#   capacity: number of (8x8 tile, Gaussian) pairs the blocks are sized for; learned from the pair counts seen so far
# ----------------------------------------------------------------------------------------------------------------
class _DeviceState:
    def __init__(self):
        self.scratch = None
        self.capacity = 1 << 16
        self.sizes = {}

    def bytes_for(self, N, H, W, cap):
        key = (N, H, W, cap)
        v = self.sizes.get(key)
        if v is None:
            lib = nat.lib()
            v = (lib.sgr_saved_bytes(N, H, W, cap), lib.sgr_scratch_bytes(N, H, W, cap))
            if len(self.sizes) > 256:
                self.sizes.clear()
            self.sizes[key] = v
        return v


_states = {}
ASYNC = os.environ.get("SPLAT_RASTER_ASYNC", "0") == "1"   # no host sync in forward; see check_overflow()
_pending_overflow_checks = []


def _state(device):
    st = _states.get(device)
    if st is None:
        st = _states[device] = _DeviceState()
    return st


def _f32c(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"diff_gaussian_rasterization (MI355X build): `{name}` must be a GPU tensor; there is no CPU path")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _empty_to_none(t):
    return None if (t is None or t.numel() == 0) else t


def _settings_struct(rs, N, M):
    s = nat.SgrSettings()
    s.num_gaussians = N
    s.image_height = int(rs.image_height)
    s.image_width = int(rs.image_width)
    s.sh_degree = int(rs.sh_degree)
    s.sh_coeffs = M
    s.tanfovx = float(rs.tanfovx)
    s.tanfovy = float(rs.tanfovy)
    s.scale_modifier = float(rs.scale_modifier)
    s.prefiltered = int(bool(rs.prefiltered))
    s.debug = int(bool(rs.debug))
    return s


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                raster_settings):
        lib = nat.lib()
        rs = raster_settings
        means3D = _f32c(means3D, "means3D")
        dev = means3D.device
        N = means3D.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        sh = _f32c(_empty_to_none(sh), "shs")
        colors_precomp = _f32c(_empty_to_none(colors_precomp), "colors_precomp")
        opacities = _f32c(opacities, "opacities")
        scales = _f32c(_empty_to_none(scales), "scales")
        rotations = _f32c(_empty_to_none(rotations), "rotations")
        cov3Ds_precomp = _f32c(_empty_to_none(cov3Ds_precomp), "cov3D_precomp")
        bg = _f32c(rs.bg, "bg")
        view = _f32c(rs.viewmatrix, "viewmatrix")
        proj = _f32c(rs.projmatrix, "projmatrix")
        praw = _f32c(rs.projmatrix_raw, "projmatrix_raw")
        campos = _f32c(rs.campos, "campos")
        M = sh.shape[1] if sh is not None else 0

        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        opac = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((N,), dtype=torch.int32, device=dev)
        n_touched = torch.empty((N,), dtype=torch.int32, device=dev)

        s = _settings_struct(rs, N, M)
        s.bg, s.viewmatrix, s.projmatrix, s.projmatrix_raw, s.campos = (
            bg.data_ptr(), view.data_ptr(), proj.data_ptr(), praw.data_ptr(), campos.data_ptr())
        inp = nat.SgrInputs(nat.ptr(means3D), nat.ptr(opacities), nat.ptr(sh), nat.ptr(colors_precomp), nat.ptr(scales),
                            nat.ptr(rotations), nat.ptr(cov3Ds_precomp))
        out = nat.SgrOutputs(color.data_ptr(), depth.data_ptr(), opac.data_ptr(), radii.data_ptr(), n_touched.data_ptr())
        st = _state(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        R = C.c_int64(0)
        with torch.cuda.device(dev):
            while True:
                cap = st.capacity
                saved_bytes, scratch_bytes = st.bytes_for(N, H, W, cap)
                if st.scratch is None or st.scratch.numel() < scratch_bytes:
                    st.scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
                saved = torch.empty(saved_bytes, dtype=torch.uint8, device=dev)
                ws = nat.SgrWorkspace(saved.data_ptr(), saved_bytes, st.scratch.data_ptr(), st.scratch.numel(), cap)
                rc = lib.sgr_forward(C.byref(s), C.byref(inp), C.byref(out), C.byref(ws),
                                     None if ASYNC else C.byref(R), stream)
                if rc == nat.SGR_ERR_CAPACITY:
                    st.capacity = int(R.value * 1.25) + 1024
                    continue
                nat.check(rc, "sgr_forward")
                break
        if ASYNC:
            _pending_overflow_checks.append((saved, dev))
        elif R.value * 2 > st.capacity:
            st.capacity = int(R.value * 2)      # head-room so that growth rarely forces a retry
        ctx.raster_settings = rs
        ctx.capacity = cap
        ctx.has_theta = theta is not None and theta.numel() == 3
        ctx.has_rho = rho is not None and rho.numel() == 3
        ctx.pose_like = (theta if ctx.has_theta else None, rho if ctx.has_rho else None)
        ctx.keep = (bg, view, proj, praw, campos)
        ctx.save_for_backward(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, radii, saved)
        ctx.mark_non_differentiable(radii, n_touched)
        return color, radii, depth, opac, n_touched

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_opacity, grad_n_touched):
        lib = nat.lib()
        rs = ctx.raster_settings
        means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, radii, saved = ctx.saved_tensors
        bg, view, proj, praw, campos = ctx.keep
        dev = means3D.device
        N = means3D.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        M = sh.shape[1] if sh is not None else 0
        grad_color = (torch.zeros((3, H, W), dtype=torch.float32, device=dev) if grad_color is None
                      else grad_color.float().contiguous())
        grad_depth = None if grad_depth is None else grad_depth.float().contiguous()
        # grad of the `opacity` image is ignored exactly like upstream (the reference never differentiates it:
        # slam_utils.py:71-77 / :108-119)

        def new(shape):
            return torch.empty(shape, dtype=torch.float32, device=dev)

        d_means3D, d_means2D, d_opac = new((N, 3)), new((N, 3)), new((N, 1))
        d_sh = new((N, M, 3)) if sh is not None else None
        d_col = new((N, 3)) if colors_precomp is not None else None
        d_scales = new((N, 3)) if scales is not None else None
        d_rot = new((N, 4)) if rotations is not None else None
        d_cov = new((N, 6)) if cov3Ds_precomp is not None else None
        d_tau = new((6,))

        s = _settings_struct(rs, N, M)
        s.bg, s.viewmatrix, s.projmatrix, s.projmatrix_raw, s.campos = (
            bg.data_ptr(), view.data_ptr(), proj.data_ptr(), praw.data_ptr(), campos.data_ptr())
        inp = nat.SgrInputs(nat.ptr(means3D), nat.ptr(opacities), nat.ptr(sh), nat.ptr(colors_precomp), nat.ptr(scales),
                            nat.ptr(rotations), nat.ptr(cov3Ds_precomp))
        go = nat.SgrGradOutputs(grad_color.data_ptr(), nat.ptr(grad_depth))
        gi = nat.SgrGradInputs(nat.ptr(d_means3D), nat.ptr(d_means2D), nat.ptr(d_opac), nat.ptr(d_sh), nat.ptr(d_col),
                               nat.ptr(d_scales), nat.ptr(d_rot), nat.ptr(d_cov), nat.ptr(d_tau))
        st = _state(dev)
        cap = ctx.capacity
        _, scratch_bytes = st.bytes_for(N, H, W, cap)
        with torch.cuda.device(dev):
            if st.scratch is None or st.scratch.numel() < scratch_bytes:
                st.scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
            ws = nat.SgrWorkspace(saved.data_ptr(), saved.numel(), st.scratch.data_ptr(), st.scratch.numel(), cap)
            nat.check(lib.sgr_backward(C.byref(s), C.byref(inp), radii.data_ptr(), C.byref(go), C.byref(gi), C.byref(ws),
                                       torch.cuda.current_stream(dev).cuda_stream), "sgr_backward")
        theta_like, rho_like = ctx.pose_like
        g_rho = d_tau[:3].reshape(rho_like.shape).to(rho_like.dtype) if ctx.has_rho else None
        g_theta = d_tau[3:].reshape(theta_like.shape).to(theta_like.dtype) if ctx.has_theta else None
        return (d_means3D, d_means2D, d_sh, d_col, d_opac.reshape(opacities.shape), d_scales, d_rot, d_cov,
                g_theta, g_rho, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, theta, rho, raster_settings)


def check_overflow():
    """ASYNC mode only: synchronises once and raises if any forward since the last call dropped pairs."""
    lib = nat.lib()
    bad = 0
    worst = 0
    while _pending_overflow_checks:
        saved, dev = _pending_overflow_checks.pop()
        R, ov = C.c_int64(0), C.c_int32(0)
        with torch.cuda.device(dev):
            nat.check(lib.sgr_query(saved.data_ptr(), C.byref(R), C.byref(ov), torch.cuda.current_stream(dev).cuda_stream),
                      "sgr_query")
        st = _state(dev)
        if R.value * 2 > st.capacity:
            st.capacity = int(R.value * 2)
        if ov.value:
            bad += 1
            worst = max(worst, R.value)
    if bad:
        raise RuntimeError(f"{bad} rasterizer forward(s) overflowed the pair capacity (needed {worst}); "
                           "capacity has been raised, re-run the step")


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Frustum test of the upstream module (p_view.z > near plane); unused by Splat-SLAM."""
        with torch.no_grad():
            vm = self.raster_settings.viewmatrix
            z = positions @ vm[:3, 2] + vm[3, 2]
            return z > 0.001

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, theta=None, rho=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   theta, rho, rs)
