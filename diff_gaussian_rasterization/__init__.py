"""Drop-in for the `diff_gaussian_rasterization` package that Splat-SLAM imports at
/root/reference/thirdparty/gaussian_splatting/gaussian_renderer/__init__.py:15-18 -- same module name, same two
public symbols, same call signature (:58-74, :130-141), same 5-tuple result (:130) -- backed by the hand-written
gfx950 kernels in libsplat_hip.so through the C ABI of include/splat_hip.h.

The upstream package (rmurai0610/diff-gaussian-rasterization-w-pose @ 43e21bf) is CUDA; this one is MI355X only.
There is no CPU / eager fallback: CPU tensors or a missing library raise.
"""
# flake8: noqa: E501
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
#   saved   : one block per forward call (up to 12 forwards are outstanding before one backward, mapper.py:426-490),
#             leased from a per-shape pool and handed back when autograd drops the graph node -- a block that went through
#             a forward has clean per-tile counters, so the library skips its zeroing launch
#   scratch : one growing block per device, shared by all calls on the stream
#   capacity: number of (8x8 tile, Gaussian) pairs the blocks are sized for = max(1 Mi, 4x the largest pair count seen so
#             far): 288 GB of HBM make that head-room free (4 B + 56 B of scratch per pair), and the pair count of a map
#             drifts by per cents between calls while different cameras of one map differ by small factors.
# No host synchronisation in the steady state: the pair count R of a forward comes back through a 64-byte asynchronous copy
# of the saved block's header into a pinned ring, read at the start of a LATER forward.  Only the first forward of a new MAP
# (another `means3D` storage or another N: a densification, a different model) waits for its R.  A forward that still finds
# the capacity too small drops pairs and the next call raises.  SPLAT_RASTER_SYNC=1 makes every forward wait (upstream's
# behaviour: never drops anything).
# ----------------------------------------------------------------------------------------------------------------
SYNC = os.environ.get("SPLAT_RASTER_SYNC", "0") == "1"
_RING = 64
_SENTINEL = 0xFFFFFFFF
_CAP_FLOOR = 1 << 20
_CAP_FACTOR = 4


class _Lease:
    """Returns a saved block to its pool when the autograd node that holds it dies."""
    __slots__ = ("pool", "block")

    def __init__(self, pool, block):
        self.pool, self.block = pool, block

    def __del__(self):
        if len(self.pool) < 32:
            self.pool.append(self.block)


class _DeviceState:
    def __init__(self, dev):
        self.dev = dev
        self.scratch = None
        self.capacity = _CAP_FLOOR
        self.sizes = {}
        self.pools = {}
        self.last_map = None
        self.last_pairs = 0               # pair count of the most recent forward whose header has arrived
        self.ring = torch.empty((_RING, 16), dtype=torch.int32, pin_memory=True)   # headers of recent forwards
        self.ring_np = self.ring.numpy().view("uint32")
        self.ring_ptr = self.ring.data_ptr()
        self.pending = []          # (slot, capacity the forward ran with)
        self.next_slot = 0
        self.overflowed = 0

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

    def lease(self, N, H, W, cap, saved_bytes):
        key = (N, H, W, cap)
        pool = self.pools.get(key)
        if pool is None:
            if len(self.pools) > 8:                # map size / capacity changed a few times: forget the old shapes
                self.pools.clear()
            pool = self.pools[key] = []
        if pool:
            return _Lease(pool, pool.pop()), 1     # went through a forward with this layout: counters are clean
        return _Lease(pool, torch.empty(saved_bytes, dtype=torch.uint8, device=self.dev)), 0

    def drain(self, wait=False):
        """Folds the pair counts that have arrived into `capacity`; returns the number of forwards that dropped pairs."""
        bad = 0
        while self.pending:
            slot, cap = self.pending[0]
            if self.ring_np[slot, 15] == _SENTINEL:            # the copy has not landed yet
                if not wait:
                    break
                torch.cuda.current_stream(self.dev).synchronize()
            R = int(self.ring_np[slot, 0])
            if int(self.ring_np[slot, 1]) == 2:                # header.overflow == 2: a 16-bit tile counter saturated
                self.pending.pop(0)
                raise RuntimeError("more than 65280 splats on one 8x8 tile: the map has degenerated")
            self.last_pairs = R
            if _CAP_FACTOR * R > self.capacity:
                self.capacity = _CAP_FACTOR * R
            if R > cap:
                bad += 1
            self.pending.pop(0)
        self.overflowed += bad
        return bad

    def post(self, saved_ptr, cap, stream):
        if len(self.pending) >= _RING - 1:
            self.drain(wait=True)
        slot = self.next_slot
        self.next_slot = (slot + 1) % _RING
        self.ring_np[slot, 15] = _SENTINEL                      # (the header's last pad word is 0 on the device)
        nat.check(nat.lib().sgr_header_to_host(saved_ptr, self.ring_ptr + 64 * slot, stream), "sgr_header_to_host")
        self.pending.append((slot, cap))
        return slot


_states = {}


def _state(device):
    st = _states.get(device)
    if st is None:
        st = _states[device] = _DeviceState(device)
    return st


def _f32c(t, name=None):
    if t is None:
        return None
    if t.dtype is not torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _empty_to_none(t):
    return None if (t is None or t.numel() == 0) else t


def _settings_struct(rs, N, M, bg, view, proj, praw, campos):
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
    s.bg, s.viewmatrix, s.projmatrix, s.projmatrix_raw, s.campos = (
        bg.data_ptr(), view.data_ptr(), proj.data_ptr(), praw.data_ptr(), campos.data_ptr())
    return s


def _stream_of(dev):
    if dev.index is not None and dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)                 # (the reference runs the mapper on one device; stay there)
    return torch.cuda.current_stream().cuda_stream


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                raster_settings):
        lib = nat.lib()
        rs = raster_settings
        if not means3D.is_cuda:
            raise RuntimeError("diff_gaussian_rasterization (MI355X build): `means3D` must be a GPU tensor; there is no CPU path")
        means3D = _f32c(means3D)
        dev = means3D.device
        N = means3D.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        sh = _f32c(_empty_to_none(sh))
        colors_precomp = _f32c(_empty_to_none(colors_precomp))
        opacities = _f32c(opacities)
        scales = _f32c(_empty_to_none(scales))
        rotations = _f32c(_empty_to_none(rotations))
        cov3Ds_precomp = _f32c(_empty_to_none(cov3Ds_precomp))
        bg, view, proj, praw, campos = (_f32c(rs.bg), _f32c(rs.viewmatrix), _f32c(rs.projmatrix), _f32c(rs.projmatrix_raw),
                                        _f32c(rs.campos))
        if not (opacities.is_cuda and bg.is_cuda and view.is_cuda and proj.is_cuda and praw.is_cuda and campos.is_cuda):
            raise RuntimeError("diff_gaussian_rasterization (MI355X build): every tensor must live on the GPU; there is no CPU path")
        M = sh.shape[1] if sh is not None else 0
        stream = _stream_of(dev)
        st = _state(dev)
        if st.pending and st.drain():
            raise RuntimeError("an earlier rasterizer forward exceeded the (tile, Gaussian) pair capacity and dropped pairs; "
                               "the capacity has been raised -- re-run the step (SPLAT_RASTER_SYNC=1 never drops pairs)")

        # one arena per call: colour | depth | opacity, and radii | n_touched
        HW = H * W
        fbuf = torch.empty(5 * HW, dtype=torch.float32, device=dev)
        color, depth, opac = fbuf[:3 * HW].view(3, H, W), fbuf[3 * HW:4 * HW].view(1, H, W), fbuf[4 * HW:].view(1, H, W)
        ibuf = torch.empty(2 * N, dtype=torch.int32, device=dev)
        radii, n_touched = ibuf[:N], ibuf[N:]

        s = _settings_struct(rs, N, M, bg, view, proj, praw, campos)
        inp = nat.SgrInputs(nat.ptr(means3D), nat.ptr(opacities), nat.ptr(sh), nat.ptr(colors_precomp), nat.ptr(scales),
                            nat.ptr(rotations), nat.ptr(cov3Ds_precomp))
        out = nat.SgrOutputs(color.data_ptr(), depth.data_ptr(), opac.data_ptr(), radii.data_ptr(), n_touched.data_ptr())
        this_map = (N, means3D.data_ptr())
        wait = SYNC or this_map != st.last_map   # a new map: learn its pair count before trusting the capacity
        R = C.c_int64(0)
        while True:
            cap = st.capacity
            saved_bytes, scratch_bytes = st.bytes_for(N, H, W, cap)
            if st.scratch is None or st.scratch.numel() < scratch_bytes:
                st.scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
            lease, clean = st.lease(N, H, W, cap, saved_bytes)
            saved = lease.block
            ntiles = ((H + 7) // 8) * ((W + 7) // 8)
            ws = nat.SgrWorkspace(saved.data_ptr(), saved_bytes, st.scratch.data_ptr(), st.scratch.numel(), cap, clean,
                                  max(1, st.last_pairs // ntiles))
            rc = lib.sgr_forward(C.byref(s), C.byref(inp), C.byref(out), C.byref(ws), C.byref(R) if wait else None, stream)
            if rc == nat.SGR_ERR_CAPACITY:
                st.capacity = int(R.value) * _CAP_FACTOR
                lease.pool = []                   # (layout changes with the capacity: do not hand this block back)
                continue
            nat.check(rc, "sgr_forward")
            break
        if wait:
            st.last_map = this_map
            st.last_pairs = int(R.value)
            if R.value * _CAP_FACTOR > st.capacity:
                st.capacity = int(R.value) * _CAP_FACTOR
        else:
            st.post(saved.data_ptr(), cap, stream)
        ctx.raster_settings = rs
        ctx.capacity = cap
        ctx.lease = lease
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
        grad_color = torch.zeros((3, H, W), dtype=torch.float32, device=dev) if grad_color is None else _f32c(grad_color)
        grad_depth = None if grad_depth is None else _f32c(grad_depth)
        # grad of the `opacity` image is ignored exactly like upstream (the reference never differentiates it:
        # slam_utils.py:71-77 / :108-119)

        # one arena for every gradient this call returns (each is written once, in full, by the gather pass)
        widths = [3, 3, 1, 3 * M if sh is not None else 0, 3 if colors_precomp is not None else 0,
                  3 if scales is not None else 0, 4 if rotations is not None else 0, 6 if cov3Ds_precomp is not None else 0]
        arena = torch.empty(N * sum(widths) + 8, dtype=torch.float32, device=dev)
        parts, o = [], 0
        for w_ in widths:
            parts.append(arena[o:o + N * w_] if w_ else None)
            o += N * w_
        d_tau = arena[o:o + 6]
        d_means3D, d_means2D, d_opac, d_sh, d_col, d_scales, d_rot, d_cov = parts

        s = _settings_struct(rs, N, M, bg, view, proj, praw, campos)
        inp = nat.SgrInputs(nat.ptr(means3D), nat.ptr(opacities), nat.ptr(sh), nat.ptr(colors_precomp), nat.ptr(scales),
                            nat.ptr(rotations), nat.ptr(cov3Ds_precomp))
        go = nat.SgrGradOutputs(grad_color.data_ptr(), nat.ptr(grad_depth))
        gi = nat.SgrGradInputs(nat.ptr(d_means3D), nat.ptr(d_means2D), nat.ptr(d_opac), nat.ptr(d_sh), nat.ptr(d_col),
                               nat.ptr(d_scales), nat.ptr(d_rot), nat.ptr(d_cov), nat.ptr(d_tau))
        st = _state(dev)
        cap = ctx.capacity
        _, scratch_bytes = st.bytes_for(N, H, W, cap)
        stream = _stream_of(dev)
        if st.scratch is None or st.scratch.numel() < scratch_bytes:
            st.scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
        ws = nat.SgrWorkspace(saved.data_ptr(), saved.numel(), st.scratch.data_ptr(), st.scratch.numel(), cap)
        nat.check(lib.sgr_backward(C.byref(s), C.byref(inp), radii.data_ptr(), C.byref(go), C.byref(gi), C.byref(ws), stream),
                  "sgr_backward")
        theta_like, rho_like = ctx.pose_like
        g_rho = d_tau[:3].reshape(rho_like.shape).to(rho_like.dtype) if ctx.has_rho else None
        g_theta = d_tau[3:].reshape(theta_like.shape).to(theta_like.dtype) if ctx.has_theta else None
        return (d_means3D.view(N, 3), d_means2D.view(N, 3), d_sh.view(N, M, 3) if d_sh is not None else None,
                d_col.view(N, 3) if d_col is not None else None, d_opac.view(opacities.shape),
                d_scales.view(N, 3) if d_scales is not None else None, d_rot.view(N, 4) if d_rot is not None else None,
                d_cov.view(N, 6) if d_cov is not None else None, g_theta, g_rho, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, theta, rho, raster_settings)


def check_overflow():
    """Waits for the pair counts of all forwards issued so far and raises if any of them dropped pairs."""
    bad = 0
    for st in _states.values():
        bad += st.drain(wait=True)
    if bad:
        raise RuntimeError(f"{bad} rasterizer forward(s) exceeded the pair capacity; it has been raised, re-run the step")


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
