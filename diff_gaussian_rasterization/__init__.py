"""Drop-in for the `diff_gaussian_rasterization` package that Splat-SLAM imports at
/root/reference/thirdparty/gaussian_splatting/gaussian_renderer/__init__.py:15-18 -- same module name, same two
public symbols, same call signature (:58-74, :130-141), same 5-tuple result (:130) -- backed by the hand-written
gfx950 kernels in libsplat_hip.so through the C ABI of include/splat_hip.h.

The upstream package (rmurai0610/diff-gaussian-rasterization-w-pose @ 43e21bf) is CUDA; this one is MI355X only.
There is no CPU / eager fallback: CPU tensors or a missing library raise.

How a mapping iteration runs through it (src/mapper.py:426-490: up to 12 render() calls, their losses summed, ONE
loss.backward()).  Every forward is one C-ABI call (four launches).  The backward is BATCHED: the renders of one iteration
receive the same parameter tensors, so the first of them routes those tensors through an identity autograd node (the
"collector") whose aliases all of them consume.  A view's own backward node then only records the image gradients it was
handed and returns no gradient for the shared inputs; autograd runs the collector's backward exactly once, after the last
of them -- and there ONE sgr_backward_views call does the tile backward, the projection backward and the gather of all
views (one launch per stage) and returns the five summed gradients.  Per view that replaces two launches, six dense
[N, .] gradient tensors and their accumulation by the autograd engine.  Views whose inputs are not shared (the caller
recomputes its activations per render) form batches of one: same results, no saving.  SPLAT_RASTER_BATCH=0 switches the
batching off (every view returns its own dense gradients, as upstream does).

Round 4: for the reference's own call (sh_degree 0, shs [N,1,3], scales + rotations, fp32 contiguous GPU tensors) the nodes and the
workspace state machine described above run in C++ (csrc/dgr_native.cpp -> _dgr.so, built by __graft_entry__.build(): libtorch
autograd nodes over the same C ABI): a forward is ONE call into the extension and the backward never re-enters the interpreter
(a 12-view iteration of the reference-shaped loop: 3.4 -> see BASELINE.md ms of host time).  This file keeps every other input
combination (precomputed colours / covariances, SH degrees 1-3, SPLAT_RASTER_BATCH=0) and is the specification the C++ half was
written from; SPLAT_RASTER_NATIVE=0 runs everything through it.
"""
# flake8: noqa: E501
from typing import NamedTuple

import collections
import ctypes as C
import os
import time
import weakref

import torch
import torch.nn as nn

from splat_slam_amd import _native as nat

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "native_extension"]


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
#   scratch : a pool of blocks per device: block 0 serves every forward (and an unbatched backward); the batched backward
#             hands view v block v (its partial slots and gradient records must outlive the launch of the other views)
#   capacity: number of (8x8 tile, Gaussian) pairs the blocks are sized for = max(1 Mi, 8 x N, 512 x tiles, 4 x the largest pair
#             count seen so far): 288 GB of HBM make that head-room free (4 B of saved and 72 B of scratch per pair), a map's pair
#             count drifts by per cents between calls, cameras of one map differ by small factors, and a view of a SLAM map with
#             more than 8 bins per Gaussian or lists of 512 on EVERY tile does not occur.
# Upstream never drops a pair (it sizes its buffers inside the call, with a host synchronisation per forward).  Here:
#   * the first forward of a new map (another N or another `means3D` storage) waits for its pair count R like upstream does;
#   * later forwards are asynchronous: R comes back through a 64-byte copy of the saved block's header into a pinned ring;
#   * the headers of all forwards of an iteration are checked INSIDE loss.backward(), before any gradient is produced: a
#     forward that exceeded the capacity (its image was rendered from truncated lists) raises there -- the optimiser has not
#     stepped, nothing was consumed silently -- and the capacity is raised for the re-run.  Forward-only renders are checked by
#     the next call.  SPLAT_RASTER_SYNC=1 makes every forward wait for R and re-run in place: upstream's guarantee, at a
#     host synchronisation per forward.
# ----------------------------------------------------------------------------------------------------------------
SYNC = os.environ.get("SPLAT_RASTER_SYNC", "0") == "1"
BATCH = os.environ.get("SPLAT_RASTER_BATCH", "1") != "0"
# The camera-pose gradients (theta, rho: 3 floats each per view) of a batch are written into `.grad` by the collector -- set, or
# added in ONE multi-tensor launch -- instead of travelling through 2 AccumulateGrad nodes (+ 2 tiny launches) per view:
# `loss.backward()` leaves the same `.grad`s as upstream.  0: strict autograd semantics also for torch.autograd.grad / tensor
# hooks on the pose deltas (the nodes then return the pose gradients themselves).  The C++ nodes (round 4) extend the deferral to
# every hook-free LEAF whose `.grad` the mapping loop reads -- screenspace_points and the exposure parameters -- and give deferred
# leaves no autograd edge at all: 60 AccumulateGrad visits less per 12-view backward pass.
DEFER_POSE_GRADS = os.environ.get("SPLAT_RASTER_DEFER_POSE_GRADS", "1") != "0"
_RING = 64
_SENTINEL = 0xFFFFFFFF
_CAP_FLOOR = 1 << 20
_CAP_FACTOR = 4
_OVERFLOW_MSG = ("a rasterizer forward exceeded the (tile, Gaussian) pair capacity: its image was rendered from truncated tile lists. "
                 "No gradient has been produced and the capacity has been raised -- re-run the iteration "
                 "(SPLAT_RASTER_SYNC=1 sizes every forward synchronously and never drops pairs)")


class CapacityOverflow(RuntimeError):
    """A forward dropped (tile, Gaussian) pairs and nothing could repair it: no gradient was produced, the capacity has been raised."""


def is_capacity_overflow(exc):
    """True for the error both node implementations raise when a truncated forward cannot be repaired (the C++ half raises it as a
    plain RuntimeError with the same text).  Callers that retry an iteration ask THIS, not the message."""
    return isinstance(exc, CapacityOverflow) or (isinstance(exc, RuntimeError) and str(exc).startswith(_OVERFLOW_MSG[:60]))


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
        self.scratch = []                 # blocks of the current scratch size
        self.scratch_bytes = 0
        self.capacity = _CAP_FLOOR
        self.sizes = {}
        self.pools = {}
        self.last_map = None
        self.last_pairs = 0               # pair count of the most recent forward whose header has arrived
        self.longest_list = 0             # longest per-tile list of recent forwards (header word 10): picks the sort build
        self.ring = torch.empty((_RING, 16), dtype=torch.int32, pin_memory=True)   # headers of recent forwards
        self.ring_np = self.ring.numpy().view("uint32")
        self.ring_ptr = self.ring.data_ptr()
        self.pending = []          # (slot, capacity the forward ran with)
        self.next_slot = 0
        self.overflowed = 0
        self.unreported = 0        # forwards found truncated that no caller has been told about yet
        self.batch = None          # the batch forwards currently join (see _Batch)

    def floor_for(self, N, ntiles):
        return max(_CAP_FLOOR, 8 * N, 512 * ntiles)

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

    def scratch_block(self, k, nbytes):
        """Scratch block k (0 = the one every forward uses), at least `nbytes` large."""
        if nbytes > self.scratch_bytes:
            self.scratch, self.scratch_bytes = [], nbytes
        while len(self.scratch) <= k:
            self.scratch.append(torch.empty(self.scratch_bytes, dtype=torch.uint8, device=self.dev))
        return self.scratch[k]

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
                # the copy was enqueued right behind its forward: it lands long before later work on the stream finishes, so
                # poll the pinned word for a while before falling back to a stream synchronisation
                t_end = time.perf_counter() + 200e-6
                while self.ring_np[slot, 15] == _SENTINEL and time.perf_counter() < t_end:
                    pass
                if self.ring_np[slot, 15] == _SENTINEL:
                    torch.cuda.current_stream(self.dev).synchronize()
            R = int(self.ring_np[slot, 0])
            if int(self.ring_np[slot, 1]) == 2:                # header.overflow == 2: a 16-bit tile counter saturated
                self.pending.pop(0)
                raise RuntimeError("more than 65280 splats on one 8x8 tile: the map has degenerated")
            self.last_pairs = R
            self.longest_list = max(int(self.ring_np[slot, 10]), (self.longest_list * 7) // 8)     # (decays when lists shrink)
            if _CAP_FACTOR * R > self.capacity:
                self.capacity = _CAP_FACTOR * R
            if R > cap:
                bad += 1
            self.pending.pop(0)
        self.overflowed += bad
        self.unreported += bad
        return bad

    def report(self, wait=False):
        """Raises if a forward since the last report dropped pairs (checked without waiting unless `wait`)."""
        if self.pending:
            self.drain(wait)
        if self.unreported:
            self.unreported = 0
            raise CapacityOverflow(_OVERFLOW_MSG)

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

# ---- the C++ half ----------------------------------------------------------------------------------------------------------------
NATIVE = os.environ.get("SPLAT_RASTER_NATIVE", "1") != "0"
# Renders whose parameter tensors are NEW objects with the same autograd provenance (the unmodified reference getters build their
# activations per call: scene/gaussian_model.py:76-101) join the batch of the iteration (C++ nodes; Batch::matches in dgr_native.cpp).
# 0: tensor identity only, i.e. one backward pass per render under those getters.
PROVENANCE = os.environ.get("SPLAT_RASTER_PROVENANCE", "1") != "0"
_ext, _ext_tried = None, False


def set_provenance_batching(on):
    """Run-time switch of PROVENANCE (tests / A-B measurements)."""
    global PROVENANCE
    PROVENANCE = bool(on)
    if _ext is not None:
        _ext.set_provenance(PROVENANCE)


def native_extension():
    """diff_gaussian_rasterization/_dgr.so (C++ autograd nodes + workspace state; splat_slam_amd.build.build_dropin_ext) or None when
    it has not been built or SPLAT_RASTER_NATIVE=0.  A library that exists but does not load raises: a broken build must not
    silently run the slower path."""
    global _ext, _ext_tried
    if not NATIVE:
        return None
    if not _ext_tried:
        _ext_tried = True
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dgr.so")
        if os.path.exists(path):
            import importlib.util
            nat.lib()                            # libsplat_hip.so first (same object the extension links)
            spec = importlib.util.spec_from_file_location("diff_gaussian_rasterization._dgr", path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            if mod.abi_version() != nat.lib().sgr_abi_version():
                raise ImportError("diff_gaussian_rasterization/_dgr.so was built against another libsplat_hip.so ABI: rebuild")
            _ext = mod
            mod.set_provenance(PROVENANCE)
        else:
            import warnings
            warnings.warn("diff_gaussian_rasterization/_dgr.so is not built (python -m splat_slam_amd.build): the rasterizer's autograd "
                          "nodes run in Python (same HIP kernels, ~3x the host time per mapping iteration)", RuntimeWarning)
    return _ext


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
    return nat.SgrSettings(N, int(rs.image_height), int(rs.image_width), int(rs.sh_degree), M, float(rs.tanfovx), float(rs.tanfovy),
                           float(rs.scale_modifier), int(bool(rs.prefiltered)), int(bool(rs.debug)), bg.data_ptr(), view.data_ptr(),
                           proj.data_ptr(), praw.data_ptr(), campos.data_ptr())


def _stream_of(dev):
    if dev.index is not None and dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)                 # (the reference runs the mapper on one device; stay there)
    return torch.cuda.current_stream().cuda_stream


# ----------------------------------------------------------------------------------------------------------------
# batched backward
# ----------------------------------------------------------------------------------------------------------------
class _ViewRecord:
    """What the collector needs from one forward to run its backward later."""
    __slots__ = ("settings", "keep", "radii", "lease", "cap", "saved_bytes", "H", "W", "grad_color", "grad_depth", "arena",
                 "means2D", "theta", "rho", "armed", "strict_pose", "__weakref__")


class _Batch:
    """The forwards that share one set of input tensors (the renders of a mapping iteration).  `inputs` holds the caller's
    tensors (identity decides membership: objects, not id()s), `alias` the collector's outputs the forwards consume.
    A record belongs to its forward's autograd node (a render whose outputs are dropped frees its saved block right away,
    however long the batch lives: an evaluation loop renders thousands of views of one parameter set); the batch only
    holds the records whose backward has been called (`armed`) until the collector has used them."""
    __slots__ = ("inputs", "versions", "alias", "armed", "closed", "dev", "N", "M", "__weakref__")

    def matches(self, tensors):
        return (not self.closed and len(tensors) == len(self.inputs) and all(a is b for a, b in zip(self.inputs, tensors))
                and all(t._version == v for t, v in zip(tensors, self.versions)))


class _Collect(torch.autograd.Function):
    """Identity on (means3D, sh, opacities, scales, rotations).  Its backward runs once per backward pass, after every view
    that consumed its outputs has recorded its image gradients: the batched backward of all of them happens here."""

    @staticmethod
    def forward(ctx, batch, means3D, sh, opacities, scales, rotations):
        ctx.batch_ref = weakref.ref(batch)     # (weak: batch -> aliases -> this node -> ctx must not close a cycle that pins HBM)
        ctx.set_materialize_grads(False)
        return means3D.detach(), sh.detach(), opacities.detach(), scales.detach(), rotations.detach()

    @staticmethod
    def backward(ctx, g_means3D, g_sh, g_opac, g_scales, g_rot):
        batch = ctx.batch_ref()
        if batch is None:
            raise RuntimeError("diff_gaussian_rasterization: the renders of this backward pass are gone")
        grads = _batched_backward(batch)
        extra = (g_means3D, g_sh, g_opac, g_scales, g_rot)      # (somebody else differentiated through the aliases: rare)
        out = tuple(g if e is None else g + e for g, e in zip(grads, extra))
        return (None,) + out


_PROF = collections.Counter() if os.environ.get("SPLAT_RASTER_PROF") else None      # phase timers of the collector (scripts/profile_dropin.py)


def _tick(name, t0):
    t1 = time.perf_counter()
    _PROF[name] += t1 - t0
    return t1


def _batched_backward(batch):
    t_ = time.perf_counter() if _PROF is not None else 0.0
    if batch.closed:
        raise RuntimeError("diff_gaussian_rasterization: backward through the renders of this parameter set a second time "
                           "(their workspaces were released by the first backward; render again, or set SPLAT_RASTER_BATCH=0)")
    batch.closed = True
    if any(t._version != v for t, v in zip(batch.inputs, batch.versions)):
        # (what autograd's saved-tensor check says for upstream's save_for_backward; the batch holds aliases, not saved tensors)
        raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace operation: "
                           "the Gaussian parameters handed to the rasterizer changed between render and backward")
    lib = nat.lib()
    dev = batch.dev
    st = _state(dev)
    if st.batch is batch:
        st.batch = None
    views, batch.armed = batch.armed, []
    means3D, sh, opacities, scales, rotations = batch.alias
    N, M = batch.N, batch.M
    # every forward of this iteration has its header on the way: make sure none of them dropped pairs BEFORE producing gradients
    st.report(wait=True)
    if _PROF is not None:
        t_ = _tick("collector: wait for the forwards' headers", t_)
    # ONE arena for the five summed gradients (each written in full by the gather pass)
    widths = (3, 3 * M, 1, 3, 4)
    arena = torch.empty(N * sum(widths), dtype=torch.float32, device=dev)
    parts, o = [], 0
    for w_ in widths:
        parts.append(arena[o:o + N * w_])
        o += N * w_
    d_means3D, d_sh, d_opac, d_scales, d_rot = parts
    stream = _stream_of(dev)
    inp = nat.SgrInputs(means3D.data_ptr(), opacities.data_ptr(), sh.data_ptr(), None, scales.data_ptr(), rotations.data_ptr(), None)
    gi = nat.SgrGradInputs(d_means3D.data_ptr(), None, d_opac.data_ptr(), d_sh.data_ptr(), None, d_scales.data_ptr(), d_rot.data_ptr(),
                           None, None)
    nv = len(views)
    keep, items = [], []
    dtau = torch.zeros((max(1, nv), 6), dtype=torch.float32, device=dev)       # (rho[3], theta[3]) of every view
    dtau_ptr = dtau.data_ptr()
    for k, r in enumerate(views):
        _, scratch_bytes = st.bytes_for(N, r.H, r.W, r.cap)
        sc = st.scratch_block(k, scratch_bytes)
        keep.append(sc)
        base = r.arena.data_ptr()
        items.append(nat.SgrBackwardView(r.settings, r.radii.data_ptr(),
                                         nat.SgrWorkspace(r.lease.block.data_ptr(), r.saved_bytes, sc.data_ptr(), sc.numel(), r.cap, 0, 0),
                                         r.grad_color.data_ptr(), None if r.grad_depth is None else r.grad_depth.data_ptr(), base,
                                         (base + 12 * N) if r.strict_pose else (dtau_ptr + 24 * k)))
    if _PROF is not None:
        t_ = _tick("collector: buffers + structs", t_)
    if nv == 0:
        arena.zero_()
    else:
        arr = (nat.SgrBackwardView * nv)(*items)
        nat.check(lib.sgr_backward_views(nv, arr, C.byref(inp), C.byref(gi), stream), "sgr_backward_views")
    if _PROF is not None:
        t_ = _tick("collector: sgr_backward_views (launches)", t_)
    # the per-view gradients (means2D; pose in strict mode) were RETURNED by the views' own backward nodes as zero tensors before
    # this launch filled them.  Where autograd kept that very tensor as `.grad` (a fresh leaf: it steals a gradient nobody else
    # references) the values are in place now; where it copied or accumulated (an existing `.grad`, a retained non-leaf), the
    # copy holds zeros + whatever was there before: add the values (all of them in one multi-tensor launch).
    fix_g, fix_v = [], []
    for k, r in enumerate(views):
        base = r.arena.data_ptr()
        todo = ((r.means2D, 0, 3 * N), (r.rho, 3 * N, 3), (r.theta, 3 * N + 3, 3)) if r.strict_pose else ((r.means2D, 0, 3 * N),)
        for p, off, n in todo:
            if p is None:
                continue
            g = p.grad
            if g is not None and g.data_ptr() != base + 4 * off:
                fix_g.append(g)
                fix_v.append(r.arena[off:off + n].view(g.shape))
        if not r.strict_pose:          # deferred pose gradients: this function IS their accumulation step
            for p, lo in ((r.rho, 0), (r.theta, 3)):
                if p is None:
                    continue
                v = dtau[k, lo:lo + 3].view(p.shape)
                if p.grad is None:
                    p.grad = v
                else:
                    fix_g.append(p.grad)
                    fix_v.append(v)
    if fix_g:
        torch._foreach_add_(fix_g, fix_v)
    if _PROF is not None:
        _tick("collector: per-view leaf gradients", t_)
    return d_means3D.view(means3D.shape), d_sh.view(sh.shape), d_opac.view(opacities.shape), d_scales.view(scales.shape), d_rot.view(rotations.shape)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                raster_settings, batch=None):
        lib = nat.lib()
        rs = raster_settings
        if not means3D.is_cuda:
            raise RuntimeError("diff_gaussian_rasterization (MI355X build): `means3D` must be a GPU tensor; there is no CPU path")
        if batch is None:            # (the shared inputs of a batch were checked by _batchable: fp32, contiguous, on the GPU)
            means3D = _f32c(means3D)
            sh = _f32c(_empty_to_none(sh))
            colors_precomp = _f32c(_empty_to_none(colors_precomp))
            opacities = _f32c(opacities)
            scales = _f32c(_empty_to_none(scales))
            rotations = _f32c(_empty_to_none(rotations))
            cov3Ds_precomp = _f32c(_empty_to_none(cov3Ds_precomp))
        dev = means3D.device
        N = means3D.shape[0]
        H, W = int(rs.image_height), int(rs.image_width)
        bg, view, proj, praw, campos = (_f32c(rs.bg), _f32c(rs.viewmatrix), _f32c(rs.projmatrix), _f32c(rs.projmatrix_raw),
                                        _f32c(rs.campos))
        if not (opacities.is_cuda and bg.is_cuda and view.is_cuda and proj.is_cuda and praw.is_cuda and campos.is_cuda):
            raise RuntimeError("diff_gaussian_rasterization (MI355X build): every tensor must live on the GPU; there is no CPU path")
        M = sh.shape[1] if sh is not None else 0
        stream = _stream_of(dev)
        st = _state(dev)
        st.report()          # (headers that have landed; the forwards of an iteration are all waited for inside their backward)

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
        ntiles = ((H + 7) // 8) * ((W + 7) // 8)
        floor = st.floor_for(N, ntiles)
        if st.capacity < floor:
            st.capacity = floor
        R = C.c_int64(0)
        while True:
            cap = st.capacity
            saved_bytes, scratch_bytes = st.bytes_for(N, H, W, cap)
            scratch = st.scratch_block(0, scratch_bytes)
            lease, clean = st.lease(N, H, W, cap, saved_bytes)
            saved = lease.block
            ws = nat.SgrWorkspace(saved.data_ptr(), saved_bytes, scratch.data_ptr(), scratch.numel(), cap, clean, st.longest_list)
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
        st.post(saved.data_ptr(), cap, stream)       # (also after a synchronous forward: the header carries the longest list)
        ctx.raster_settings = rs
        ctx.has_theta = theta is not None and theta.numel() == 3
        ctx.has_rho = rho is not None and rho.numel() == 3
        ctx.mark_non_differentiable(radii, n_touched)
        if batch is not None:
            r = _ViewRecord()
            r.settings, r.keep, r.radii, r.lease, r.cap, r.saved_bytes, r.H, r.W = s, (bg, view, proj, praw, campos), radii, lease, cap, saved_bytes, H, W
            r.grad_color = r.grad_depth = r.arena = None
            r.means2D = means2D if (means2D is not None and means2D.requires_grad) else None
            r.theta = theta if (ctx.has_theta and theta.requires_grad) else None
            r.rho = rho if (ctx.has_rho and rho.requires_grad) else None
            r.armed = False
            r.strict_pose = not DEFER_POSE_GRADS
            ctx.record, ctx.batch = r, batch
            ctx.pose_like = (theta if ctx.has_theta else None, rho if ctx.has_rho else None)
            return color, radii, depth, opac, n_touched
        ctx.record = None
        ctx.capacity = cap
        ctx.lease = lease
        ctx.pose_like = (theta if ctx.has_theta else None, rho if ctx.has_rho else None)
        ctx.keep = (bg, view, proj, praw, campos)
        ctx.save_for_backward(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, radii, saved)
        return color, radii, depth, opac, n_touched

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_opacity, grad_n_touched):
        rs = ctx.raster_settings
        H, W = int(rs.image_height), int(rs.image_width)
        r = ctx.record
        if r is not None:
            # batched: record the image gradients, hand autograd zero tensors for this view's own leaves (filled by the collector)
            batch = ctx.batch
            if batch.closed:
                raise RuntimeError("diff_gaussian_rasterization: backward through a render whose batch has already run its backward "
                                   "(render again, or set SPLAT_RASTER_BATCH=0 for independent per-view backward passes)")
            dev, N = batch.dev, batch.N
            r.grad_color = torch.zeros((3, H, W), dtype=torch.float32, device=dev) if grad_color is None else _f32c(grad_color)
            r.grad_depth = None if grad_depth is None else _f32c(grad_depth)
            # grad of the `opacity` image is ignored exactly like upstream (the reference never differentiates it:
            # slam_utils.py:71-77 / :108-119)
            r.arena = torch.zeros(3 * N + 6, dtype=torch.float32, device=dev)        # dL/dmeans2D [N,3] | rho [3] | theta [3]
            if not r.armed:
                r.armed = True
                batch.armed.append(r)
            g_rho = g_theta = None
            if r.strict_pose:
                theta_like, rho_like = ctx.pose_like
                g_rho = r.arena[3 * N:3 * N + 3].view(rho_like.shape) if ctx.has_rho else None
                g_theta = r.arena[3 * N + 3:].view(theta_like.shape) if ctx.has_theta else None
            return (None, r.arena[:3 * N].view(N, 3), None, None, None, None, None, None, g_theta, g_rho, None, None)

        lib = nat.lib()
        means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, radii, saved = ctx.saved_tensors
        bg, view, proj, praw, campos = ctx.keep
        dev = means3D.device
        N = means3D.shape[0]
        M = sh.shape[1] if sh is not None else 0
        grad_color = torch.zeros((3, H, W), dtype=torch.float32, device=dev) if grad_color is None else _f32c(grad_color)
        grad_depth = None if grad_depth is None else _f32c(grad_depth)
        st = _state(dev)
        st.report(wait=True)

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
        cap = ctx.capacity
        _, scratch_bytes = st.bytes_for(N, H, W, cap)
        stream = _stream_of(dev)
        scratch = st.scratch_block(0, scratch_bytes)
        ws = nat.SgrWorkspace(saved.data_ptr(), saved.numel(), scratch.data_ptr(), scratch.numel(), cap)
        nat.check(lib.sgr_backward(C.byref(s), C.byref(inp), radii.data_ptr(), C.byref(go), C.byref(gi), C.byref(ws), stream),
                  "sgr_backward")
        theta_like, rho_like = ctx.pose_like
        g_rho = d_tau[:3].reshape(rho_like.shape).to(rho_like.dtype) if ctx.has_rho else None
        g_theta = d_tau[3:].reshape(theta_like.shape).to(theta_like.dtype) if ctx.has_theta else None
        return (d_means3D.view(N, 3), d_means2D.view(N, 3), d_sh.view(N, M, 3) if d_sh is not None else None,
                d_col.view(N, 3) if d_col is not None else None, d_opac.view(opacities.shape),
                d_scales.view(N, 3) if d_scales is not None else None, d_rot.view(N, 4) if d_rot is not None else None,
                d_cov.view(N, 6) if d_cov is not None else None, g_theta, g_rho, None, None)


def _batchable(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs):
    """The batched backward covers the reference's call (gaussian_renderer/__init__.py:130-141 at sh_degree 0: shs [N,1,3],
    scales + rotations, fp32 contiguous GPU tensors) when at least one shared input wants a gradient."""
    if not (BATCH and torch.is_grad_enabled()) or colors_precomp is not None or cov3Ds_precomp is not None:
        return False
    if sh is None or scales is None or rotations is None or int(rs.sh_degree) != 0 or sh.dim() != 3 or sh.shape[1] != 1:
        return False
    ok = False
    for t in (means3D, sh, opacities, scales, rotations):
        if not (t.is_cuda and t.dtype is torch.float32 and t.is_contiguous()):
            return False
        ok = ok or t.requires_grad
    return ok and means3D.shape[0] > 0


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                        raster_settings):
    ext = native_extension() if BATCH else None
    if ext is not None and _empty_to_none(colors_precomp) is None and _empty_to_none(cov3Ds_precomp) is None and means3D.is_cuda:
        rs = raster_settings
        out = ext.try_rasterize(means3D, means2D, _empty_to_none(sh), opacities, _empty_to_none(scales), _empty_to_none(rotations), theta, rho,
                                int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy), rs.bg,
                                float(rs.scale_modifier), rs.viewmatrix, rs.projmatrix, rs.projmatrix_raw, int(rs.sh_degree), rs.campos,
                                bool(rs.prefiltered), bool(rs.debug), SYNC, DEFER_POSE_GRADS)
        if out is not None:                  # (None: not the reference's call shape -- the Python nodes below take it)
            return out
    if _batchable(means3D, sh, _empty_to_none(colors_precomp), opacities, scales, rotations, _empty_to_none(cov3Ds_precomp),
                  raster_settings):
        st = _state(means3D.device)
        tensors = (means3D, sh, opacities, scales, rotations)
        b = st.batch
        if b is None or not b.matches(tensors):
            b = _Batch()
            b.inputs, b.versions = tensors, tuple(t._version for t in tensors)
            b.armed, b.closed, b.dev, b.N, b.M = [], False, means3D.device, means3D.shape[0], sh.shape[1]
            b.alias = _Collect.apply(b, *tensors)
            st.batch = b
        a = b.alias
        return _RasterizeGaussians.apply(a[0], means2D, a[1], None, a[2], a[3], a[4], None, theta, rho, raster_settings, b)
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, theta, rho, raster_settings, None)


def saved_block_of(output):
    """(saved workspace block, capacity) of the forward that produced `output` (one of the differentiable tensors a
    GaussianRasterizer call returned) -- parity tooling: sgr_query_* read the forward's own depth keys / counters from it."""
    ext = native_extension()
    if ext is not None:
        hit = ext.saved_block_of(output)
        if hit is not None:
            return hit
    fn = output.grad_fn
    r = getattr(fn, "record", None)
    if r is not None:
        return r.lease.block, r.cap
    return fn.saved_tensors[-1], fn.capacity


def check_overflow():
    """Waits for the pair counts of all forwards issued so far and raises if any of them dropped pairs."""
    for st in _states.values():
        st.report(wait=True)
    if native_extension() is not None:
        native_extension().check_overflow()


class GaussianRasterizer(nn.Module):
    """Same constructor and call as upstream.  The mapping loop builds one per render (gaussian_renderer/__init__.py:74):
    nn.Module.__init__ (a dozen dicts, ~10 us) is deferred until something needs the Module machinery."""

    def __init__(self, raster_settings):
        object.__setattr__(self, "raster_settings", raster_settings)

    def _late_init(self):
        d = object.__getattribute__(self, "__dict__")
        rs = d.pop("raster_settings", None)
        nn.Module.__init__(self)
        object.__setattr__(self, "raster_settings", rs)

    def __getattr__(self, name):           # only reached for attributes that are missing
        if "_parameters" not in object.__getattribute__(self, "__dict__"):
            self._late_init()
            return getattr(self, name)
        return nn.Module.__getattr__(self, name)

    def __setattr__(self, name, value):
        if "_parameters" not in self.__dict__:
            self._late_init()
        nn.Module.__setattr__(self, name, value)

    def __call__(self, *args, **kwargs):
        if "_parameters" in self.__dict__:
            return nn.Module.__call__(self, *args, **kwargs)       # (hooks may have been registered)
        return self.forward(*args, **kwargs)

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
