"""Autograd-free mapping loops for MI355X: the same three loops as splat_slam_amd.mapper.MappingLoop
(/root/reference/src/mapper.py:303-353, 400-568, 656-708) executed through the C ABI:

    sgr_map_run    a run of regular iterations between two densify / reset points = ONE host call
                   (map(iters=60), final_refine in chunks, initialize_map between its densification points)
    sgr_map_step   one iteration = activate -> batched views {forward, loss in the compositing epilogue, backward into the
                   gradient sinks + densification statistics} -> Adam of the five groups -> exposure Adam
                   (used around map surgery, with pose optimisation, and -- twice per iteration with one all-reduce in
                   between -- on several GPUs)

instead of ~60 eager torch kernels and an autograd graph per view.  Nothing on this path synchronises with the host
between two capacity checks: outputs, saved blocks and gradient sinks are persistent device buffers (HBM is 288 GB), a
camera's pair capacity is carried over from its last measured count.  The checks (end of every loop call, before map
surgery, every `check_every` iterations) close a TRANSACTION: a forward that exceeded its capacity is never dropped from
the optimisation -- the state is restored and the iterations since the last check are issued again at the corrected
capacity, bit-identical to a run that never overflowed (_txn_begin / _txn_do / _txn_commit below).

Numerically this is the same computation as the autograd loop (tests/test_gpu_fused.py compares gradients and parameter
trajectories; the run / step / fused-tail / multi-rank variants are bitwise identical among themselves); only the order
of fp32 additions differs from autograd (views are accumulated in place).

The reference's `map(prune=True)` pass (src/mapper.py:490-520) runs forward AND backward and returns before
optimizer.step() / zero_grad(): its gradients stay on every parameter tensor that is not replaced before the next step
(new keyframe points and map deformation replace the Gaussian tensors; exposure parameters and -- before final_refine
-- f_dc / opacity keep them) and are added to the next iteration's.  The fused loop reproduces that: the prune pass
leaves its gradients in the sinks, its isotropy term and exposure gradients are carried as "stale" state and consumed by
the first optimiser step that follows (tests/test_gpu_fused.py replays the reference fixture through it).
"""
import collections
import os
import ctypes as C

import numpy as np
import torch

from splat_slam_amd import _native as nat
from splat_slam_amd.mapper import MappingLoop
from splat_slam_amd.pose import update_pose

_GROUPS = ["xyz", "f_dc", "opacity", "scaling", "rotation"]     # order expected by sgr_gaussian_adam_step


class _Plan:
    pass


class _ViewBuffers:
    """Per-camera device buffers.  They SURVIVE changes of the map size (every keyframe appends Gaussians, every densification
    changes N): re-making ~150 cameras' buffers -- and above all their ~250 MB workspaces, whose sizes never repeat exactly, so
    the caching allocator went to hipMalloc for them -- cost a converged session 25-30 ms of host time per keyframe, during which
    the GPU idled.  Images depend on (H, W) only; the per-Gaussian int buffers and the workspaces are allocated with head-room
    and re-carved."""

    def __init__(self, H, W, N, dev):
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        self.dev = dev
        self.color, self.depth, self.opacity = f(3, H, W), f(1, H, W), f(1, H, W)
        self.d_color, self.d_depth = f(3, H, W), f(1, H, W)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.d_exp = torch.zeros(2, dtype=torch.float32, device=dev)
        self.d_tau = torch.zeros(6, dtype=torch.float32, device=dev)
        ntiles = ((H + 7) // 8) * ((W + 7) // 8)
        self.loss_scratch = torch.empty(max(1024, ntiles) * 16, dtype=torch.uint8, device=dev)   # one LossPart per 8x8 tile
        self._ibuf, self._icap = None, 0
        self.gen = 0               # bumped whenever a buffer a cached SgrMapView points into is re-allocated (ints, saved, scratch)
        self.saved = None
        self.scratch = None
        self.clean = False         # the saved block went through a forward (its per-tile counters are zero)
        self.ovf_seen = 0          # SavedHeader.overflow_events as of the last check (sticky on the device, see _apply_headers)
        self.capacity = 0
        self.gt_depth = None
        self.depth_src = None
        self.mv = None
        self.new_map(N)

    def new_map(self, N):
        """The map changed size: counts measured at the old size are hints only, cached structs are stale."""
        if N > self._icap:
            self._icap = int(N * 1.3) + 4096
            self._ibuf = torch.empty(2 * self._icap, dtype=torch.int32, device=self.dev)
            self.gen += 1
        self.radii, self.n_touched = self._ibuf[:N], self._ibuf[self._icap:self._icap + N]
        self.n = N
        self.pairs = -1            # (tile, Gaussian) pairs seen by the probe render; -1 = not probed at this map size
        self.estimated = False     # `pairs` is a carried-over estimate, not yet confirmed by a header read at this map size
        self.ran = False           # a forward has run on the saved block at THIS map size (its header is current)
        # (the cached SgrMapViews stay: FusedMappingLoop._map_view re-fills what depends on N)


class _IndexStage:
    """Row-index tensors on the device without a host wait.  torch.tensor(rows, device=...) is a pageable copy (waits for the
    stream); .pin_memory() per call allocates a fresh pinned block whenever the previous ones are still referenced by copies
    in flight -- with the GPU tens of milliseconds behind the host that is every call, ~0.7 ms each with the GPU idle behind it.
    Here: ONE pinned ring written in place (a slot is reused 1 << 15 entries later: long after its copy ran) + the last few
    index tensors kept by value (the window's rows repeat from iteration to iteration)."""
    RING = 1 << 15

    def __init__(self):
        self.ring = None
        self.pos = 0
        self.cache = collections.OrderedDict()

    def get(self, rows, device):
        key = (tuple(rows), str(device))
        hit = self.cache.get(key)
        if hit is not None:
            self.cache.move_to_end(key)
            return hit
        n = len(key[0])
        if n == 0 or n > self.RING // 8:
            return torch.tensor(list(rows), dtype=torch.long, device=device)
        if self.ring is None:
            self.ring = torch.empty(self.RING, dtype=torch.long, pin_memory=True)
        if self.pos + n > self.RING:
            self.pos = 0
        src = self.ring[self.pos:self.pos + n]
        self.pos += n
        src.copy_(torch.tensor(key[0], dtype=torch.long))
        out = src.to(device, non_blocking=True)
        self.cache[key] = out
        if len(self.cache) > 64:
            self.cache.popitem(last=False)
        return out


_INDEX_STAGE = _IndexStage()


def _index(rows, device):
    """Row indices on the device (see _IndexStage).  The result may be shared between callers: never written to."""
    return _INDEX_STAGE.get(list(rows), device)


class _Slot:
    """A workspace that belongs to no camera: pick j of every iteration of a span renders in slot j."""
    is_slot = True

    def __init__(self):
        self.saved = self.scratch = None
        self.clean = self.ran = False
        self.ovf_seen = 0
        self.capacity = 0
        self.mv = None
        self.gen = 0


class _ExposureSlab:
    """Exposure parameters (a, b) of every camera in one device slab + Adam moments, so the keyframe optimiser step
    (src/mapper.py:561, lr 0.01, torch defaults betas (0.9, 0.999), eps 1e-8) is ONE masked launch."""

    def __init__(self, device, capacity=4096):
        z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=device)
        self.param, self.grad, self.m, self.v = z(capacity, 2), z(capacity, 2), z(capacity, 2), z(capacity, 2)
        self.step = z(capacity, dt=torch.int32)
        self.active = z(capacity, dt=torch.int32)
        self.ones = torch.ones(capacity, dtype=torch.int32, device=device)
        self.stale = z(capacity, 2)        # exposure gradients a prune pass left behind (the reference never zeroes them)
        self.stale_rows = set()
        self.rows = {}                     # camera uid -> row
        self.owner = {}                    # row -> the Camera object whose exposure_a/b are views of it

    def row_of(self, cam):
        r = self.rows.get(cam.uid)
        return r if r is not None and self.owner.get(r) is cam else None

    def attach(self, cam):
        r = self.row_of(cam)
        if r is None:
            r = self.rows.get(cam.uid)             # a new Camera object with a known uid takes the row over
            if r is None:
                r = len(self.rows)
                if r >= self.param.shape[0]:
                    self._grow()
                self.rows[cam.uid] = r
            self.owner[r] = cam
            self.stale[r] = 0
            self.stale_rows.discard(r)
            with torch.no_grad():
                self.param[r, 0] = cam.exposure_a.detach()[0]
                self.param[r, 1] = cam.exposure_b.detach()[0]
            cam.exposure_a = torch.nn.Parameter(self.param[r, 0:1])     # views: the slab IS the parameter storage
            cam.exposure_b = torch.nn.Parameter(self.param[r, 1:2])
        return r

    def _grow(self):
        """Doubles the slab; Camera.exposure_a/b of attached cameras are re-bound to their rows."""
        old = self.param.shape[0]
        for name in ("param", "grad", "m", "v", "stale"):
            t = getattr(self, name)
            setattr(self, name, torch.cat([t, torch.zeros_like(t)]))
        self.step = torch.cat([self.step, torch.zeros_like(self.step)])
        self.active = torch.cat([self.active, torch.zeros_like(self.active)])
        self.ones = torch.ones(2 * old, dtype=torch.int32, device=self.param.device)
        for r, cam in self.owner.items():
            cam.exposure_a = torch.nn.Parameter(self.param[r, 0:1])
            cam.exposure_b = torch.nn.Parameter(self.param[r, 1:2])

    def keep_stale(self, rows):
        """A prune pass wrote its exposure gradients into grad[rows]; like `.grad` in the reference they are ADDED to."""
        if rows:
            idx = _index(sorted(set(rows)), self.param.device)
            self.stale[idx] += self.grad[idx]
            self.stale_rows.update(int(r) for r in rows)

    def add_stale(self, rows):
        """grad[rows] += stale[rows] for the rows an optimiser is about to step (and then zero_grad's): returns those rows."""
        hit = sorted(r for r in rows if r in self.stale_rows)
        if hit:
            idx = _index(hit, self.param.device)
            self.grad[idx] += self.stale[idx]
            self.stale[idx] = 0
            self.stale_rows.difference_update(hit)
        return hit

    def reset(self, rows):
        self.active.zero_()
        if rows:
            idx = _index(rows, self.param.device)
            self.m[idx] = 0
            self.v[idx] = 0
            self.step[idx] = 0
            self.active[idx] = 1

    def step_mask(self, lib, rows, stream):
        n = max(rows) + 1
        nat.check(lib.sgr_masked_adam(n, 2, self.param.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                      self.step.data_ptr(), self.active.data_ptr(), 0.01, 0.9, 0.999, 1e-8, stream),
                  "sgr_masked_adam")

    def step_rows(self, lib, row, n, stream):
        o = 8 * row
        nat.check(lib.sgr_masked_adam(n, 2, self.param.data_ptr() + o, self.grad.data_ptr() + o, self.m.data_ptr() + o,
                                      self.v.data_ptr() + o, self.step.data_ptr() + 4 * row, self.ones.data_ptr(), 0.01, 0.9,
                                      0.999, 1e-8, stream), "sgr_masked_adam")


def estimate_pairs(hints, uid, n_now):
    """(tile, Gaussian) pair count to size a camera's workspace by when its buffers are new (FusedMappingLoop._estimate_pairs).
    `hints`: uid -> (pairs measured, map size at that time).  The camera's own measurement scaled by the growth of the map
    (never down) x 1.25; for a camera never measured 1.5 x the largest such estimate of the others; None = nothing known."""
    scaled = lambda h: int(h[0] * max(1.0, n_now / max(1, h[1])) * 1.25) + 1024
    h = hints.get(uid)
    if h is not None:
        return scaled(h)
    if hints:
        return int(1.5 * max(scaled(x) for x in hints.values()))
    return None


class FusedMappingLoop(MappingLoop):
    def __init__(self, config, device="cuda:0", knn_fn=None, check_every=50, span_calls=True):
        super().__init__(config, device=device, fused_loss=True, knn_fn=knn_fn)
        # `ssim_loss: True` (slam_utils.py:89-98, off by default) is not a per-pixel L1: the fused tile kernel's loss epilogue does
        # not apply, and the three loops run as the autograd MappingLoop (drop-in rasterizer + torch loss) instead
        # `spherical_harmonics: True` (mapper.py:78,85: sh_degree 3, off and "not tested" upstream, splat_slam.yaml:59) gives the map a
        # [N, 15, 3] f_rest group the fused optimiser pass has no rows for: same fallback (the rasterizer itself handles SH 0-3)
        from splat_slam_amd.losses import uses_ssim
        self.autograd_fallback = uses_ssim(config["mapping"]) or bool(config["mapping"]["Training"].get("spherical_harmonics", False))
        self.lib = nat.lib()
        self.check_every = check_every
        self.fuse_tail = True            # gather + Adam + next activations in one pass (single GPU, regular iterations)
        self.span_calls = span_calls     # regular iterations between map-surgery points go through sgr_map_run
        self.overflow_events = 0
        self._views = {}
        self._acc = None          # gradient accumulators wrt activated inputs + activated copies
        self._acc_key = None
        self._acc_clean = True
        self._flat_dirty = False  # ZeRO-1 store-mode exchange: the flat sinks hold the last local sums, not zeros (see _exchange_and_adam)
        self._scratch = None
        self._since_check = 0
        # Never drop (round 5): everything enqueued since the last capacity check is a TRANSACTION -- a snapshot of the optimisation
        # state taken before its first launch plus a journal of the launches.  A check that finds a truncated forward restores the
        # snapshot, grows the capacity and re-issues the journal (_txn_commit), so a view that overflowed takes part in every step
        # after all, like upstream's rasterizer, which sizes its buffers inside the call (README.md:88-92 module).
        self._txn = None
        # True: a span whose views run on carried-over pair-count ESTIMATES sends its first iteration ahead and reads its headers back
        # before the rest is enqueued (round 3: an estimate that was short then cost one iteration, not a span of dropped views).
        # With transactions nothing is ever dropped -- a short estimate costs a replay of the span, once in tens of keyframes -- and the
        # extra synchronisation per keyframe (GPU idle ~1 ms behind it, scripts/session_timeline.py) is the more expensive of the two.
        self.verify_estimates = os.environ.get("SPLAT_VERIFY_ESTIMATES", "0") == "1"
        self._span_cache = None    # (window, pool) SgrMapView arrays of the spans of one map() call (_span_arrays)
        self.cache_span_arrays = os.environ.get("SPLAT_SPAN_CACHE", "1") != "0"      # (0: A/B measurements of the host path)
        self.keep_structs_over_keyframes = os.environ.get("SPLAT_KEYFRAME_STRUCTS", "1") != "0"   # (0: round-5 behaviour, for A/B)
        self.snapshot_store = os.environ.get("SPLAT_SNAPSHOT_STORE", "1") != "0"                  # (0: a fresh buffer per tensor and map size, for A/B)
        self._txn_pool = None      # snapshot buffers, reused while the tensor shapes stay
        self._txn_store = None     # ... the byte store they are views of (kept with head-room across map sizes)
        self._replaying = False
        self.replayed_transactions = 0
        self._hdr_pinned = None
        self._gen = 0              # bumped whenever cached launch structs go stale (capacity, hints, buffers)
        self._pair_hint = {}       # camera uid -> (measured pair count, map size it was measured at)
        self._list_hint = {}       # camera uid -> longest per-tile list measured (header word 10): picks the tile kernels' sort build
        self._ws_bytes = {}        # (N, H, W, capacity) -> (saved bytes, scratch bytes)
        # Workspaces (~300 MB each) are not a per-keyframe resource: a map holds hundreds of keyframes, an iteration renders 12.
        # Cameras that are rendered as REGULAR views (the window, special iterations) hold a private one, taken from the least
        # recently used owner once `max_live_ws` are out; the random picks of a span share `picks_per_iter` SLOTS (SgrMapRun.pick_ws).
        self._ws_owners = collections.OrderedDict()       # id(vb) -> vb, least recently used first
        self._ws_protect = set()                          # id(vb) of the views of the call being built
        self.max_live_ws = 32
        self._slots = []
        self._exp = None
        self._exp_rows = []
        self._cap = 0
        self._proj_raw = {}
        self._plan_key = None
        self._plan_obj = None
        # multi-GPU (set_parallel): ranks of one node share the views of an iteration and exchange gradients over RCCL
        self.world, self.rank = 1, 0
        self.split_views = True      # strong scaling: view v of an iteration is rendered by rank v mod world
        self.sync = "zero1"          # "zero1": reduce-scatter + Adam on 1/world of the optimiser + all-gather; "allreduce"
        self.comm = None
        self._zero = None            # parallel.Zero1Plan + flat parameter / moment / shard buffers (sync == "zero1")
        self._replicated = 0         # > 0: every rank runs the identical iteration, no exchange (initialize_map, final_refine)
        self._acc_ids = None        # the five parameter tensors the sinks belong to (held, so that identity checks are sound)
        self._stale_iso = 0.0       # isotropy weight whose gradient a prune pass left on the current `_scaling` tensor
        self.max_pairs = 1 << 28    # a view with more (tile, Gaussian) pairs than this is a degenerate map: fail loudly, not by OOM
        self.capacity_floor = 1 << 16      # the pair capacity never goes below this (tests: a floor no view can reach = a run without overflow)

    def reset(self):
        super().reset()
        self._views, self._acc, self._acc_key, self._acc_ids, self._acc_clean = {}, None, None, None, True
        self._exp, self._exp_rows, self._cap, self._stale_iso = None, [], 0, 0.0
        self._pair_hint, self._list_hint = {}, {}
        self._plan_key = self._plan_obj = None
        self._ws_owners.clear()
        self._slots = []
        self._txn = None
        self._span_cache = None

    # ------------------------------------------------------------------------------------------------ state
    def set_parallel(self, world, rank, split_views=True, sync="zero1", comm=None):
        """Makes this loop one of `world` ranks that map the SAME keyframes of a replicated map (one process per GPU,
        torch.distributed initialised by the caller; backend "nccl" is RCCL on ROCm and rides xGMI inside a node).

        split_views  True: the views of one iteration are dealt round-robin to the ranks (the reference's iteration in
                     parallel, mapper.py:426-490: <= 12 views, so <= 12 ranks have work); False: every rank renders all the
                     views it is given (the caller hands each rank its own: a world-times larger batch per step).
        sync         "zero1": gradients are reduce-scattered, every rank runs Adam on the rows it owns (1/world of the
                     optimiser state is touched per rank) and the parameters are all-gathered -- SURVEY.md 8e;
                     "allreduce": one all-reduce of the flat gradient buffer, replicated Adam."""
        from splat_slam_amd.parallel import Comm
        assert sync in ("zero1", "allreduce")
        self.world, self.rank, self.split_views, self.sync = int(world), int(rank), bool(split_views), sync
        self.comm = comm if comm is not None else Comm()
        self._acc_key = None            # buffers are laid out per world size

    def _parallel(self):
        return self.world > 1 and self._replicated == 0

    def _local(self, cams):
        """The views of an iteration this rank renders."""
        return list(cams[self.rank::self.world]) if (self._parallel() and self.split_views) else list(cams)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _ensure_state(self):
        gm = self.gaussians
        key = gm._xyz.shape[0]
        ids = (gm._xyz, gm._features_dc, gm._opacity, gm._scaling, gm._rotation)     # the objects, compared with `is`: an id()
        same = self._acc_ids is not None and all(a is b for a, b in zip(self._acc_ids, ids))   # can be reused once its owner died
        if self._acc_key == key and same:
            return
        if self._acc_key == key and self._acc is not None:
            # same N, some tensors replaced (opacity reset, map deformation: replace_tensor_to_optimizer): the new
            # Parameters have no .grad in the reference, i.e. whatever a prune pass left for THOSE groups is gone
            changed = [k for k, (old, new) in enumerate(zip(self._acc_ids, ids)) if old is not new]
            for k in changed:
                if not self._acc_clean:
                    self._acc[_GROUPS[k]].zero_()
                if _GROUPS[k] == "scaling":
                    self._stale_iso = 0.0
            self._acc_ids = ids
            if self._zero is not None:
                self._rehome(changed)
            return
        N, dev = gm._xyz.shape[0], self.device
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        # the five gradient accumulators are views of ONE flat buffer (14 floats per Gaussian, group-major): a multi-GPU
        # step exchanges it with a single collective, no packing (parallel.Zero1Plan: rows padded so that equal shards of
        # the buffer are whole-row segments)
        from splat_slam_amd.parallel import Zero1Plan
        plan = Zero1Plan(N, self.world, self.rank)
        flat = z(plan.total)
        shapes = {"xyz": (N, 3), "f_dc": (N, 1, 3), "opacity": (N, 1), "scaling": (N, 3), "rotation": (N, 4)}
        self._acc = {"flat": flat, "act_scale": z(N, 3), "act_rot": z(N, 4), "act_opac": z(N, 1)}
        for name, shape in shapes.items():
            self._acc[name] = plan.view(flat, name, shape)
        self._zero = None
        self._acc_key, self._acc_ids = key, ids
        self._acc_clean = True     # the gradient sinks are all-zero (every Adam step leaves them so)
        self._flat_dirty = False
        self._stale_iso = 0.0      # (every tensor is new: nothing a prune pass left behind survives)
        # N changed: the per-camera buffers stay (see _ViewBuffers), cameras that left the map are dropped
        live = set(self.viewpoints)
        self._views = {uid: vb for uid, vb in self._views.items() if uid in live}
        for vb in self._views.values():
            vb.new_map(N)
        if gm.active_sh_degree != 0 or gm._features_rest.numel() != 0:
            # (only reachable with a model that was handed to the loop from outside, e.g. a PLY with SH coefficients:
            #  `spherical_harmonics: True` in the config selects the autograd loop in __init__)
            raise NotImplementedError("the fused iteration covers the reference's default sh_degree 0 (mapper.py:85); set "
                                      "mapping.Training.spherical_harmonics: True to run this map through the autograd loop")
        for g in gm.optimizer.param_groups:          # make sure Adam state exists exactly like torch would create it
            p = g["params"][0]
            st = gm.optimizer.state.get(p)
            if st is None or len(st) == 0:
                gm.optimizer.state[p] = {"step": torch.tensor(0.0), "exp_avg": torch.zeros_like(p),
                                         "exp_avg_sq": torch.zeros_like(p)}
        if self.world > 1 and self.sync == "zero1":
            self._zero = {"plan": plan, "param": z(plan.total), "m": z(plan.total), "v": z(plan.total), "shard": z(plan.shard),
                          "shapes": shapes}
            self._rehome(range(5))

    def _rehome(self, which):
        """ZeRO-1: parameters and Adam moments of the given groups become views of the flat buffers the collectives work on
        (values copied once; the tensors keep their identity, so optimiser state and plans stay valid)."""
        gm, zr = self.gaussians, self._zero
        by_name = {g["name"]: g for g in gm.optimizer.param_groups}
        with torch.no_grad():
            for k in which:
                name = _GROUPS[k]
                p = by_name[name]["params"][0]
                st = gm.optimizer.state[p]
                for buf, src, put in ((zr["param"], p.data, lambda t, p=p: setattr(p, "data", t)),
                                      (zr["m"], st["exp_avg"], lambda t, st=st: st.__setitem__("exp_avg", t)),
                                      (zr["v"], st["exp_avg_sq"], lambda t, st=st: st.__setitem__("exp_avg_sq", t))):
                    dst = zr["plan"].view(buf, name, zr["shapes"][name])
                    dst.copy_(src.reshape(dst.shape))
                    put(dst.view(src.shape))
        gm.invalidate_activations()             # (p.data was re-pointed: no _version bump)
        self._plan_key = None

    def _sync_moments(self):
        """ZeRO-1: every rank only keeps the moments of its own rows current; before anything looks at (or re-shapes) the
        whole optimiser state -- map surgery, replicated phases, the end of a map() call -- they are all-gathered."""
        if self._zero is None or not self._zero.get("stale_moments"):
            return
        zr = self._zero
        lo, hi = zr["plan"].lo, zr["plan"].hi
        self.comm.all_gather(zr["m"], zr["m"][lo:hi])
        self.comm.all_gather(zr["v"], zr["v"][lo:hi])
        zr["stale_moments"] = False

    def _clean_flat(self):
        """The sinks as every non-store-mode path expects them when `_acc_clean` says so: all-zero."""
        if self._flat_dirty:
            self._acc["flat"].zero_()
            self._flat_dirty = False

    def _exchange_and_adam(self, pl, iso_weight, skip=(), stored=False):
        """Second half of a multi-GPU iteration: the ranks' gradient sums meet, Adam steps, everybody ends with the same
        parameters and the activations of the next forward.  pl.groups carry lr / step of this iteration."""
        stream = self._stream()
        a, gm = self._acc, self.gaussians
        for k, name in enumerate(_GROUPS):
            pl.groups[k].skip = int(name in skip)
        if self._stale_iso and "scaling" not in skip:
            iso_weight, self._stale_iso = iso_weight + self._stale_iso, 0.0
        if self._zero is None:
            self.comm.all_reduce(a["flat"])
            adam_st = nat.SgrMapStep()
            C.memmove(C.byref(adam_st), C.byref(pl.step), C.sizeof(adam_st))
            # optimiser-only step: its Adam pass also writes the activations of the updated parameters
            adam_st.num_views, adam_st.views, adam_st.adam_groups, adam_st.exp_rows = 0, None, pl.groups, 0
            adam_st.iso_weight, adam_st.forward_only, adam_st.grads_clean = float(iso_weight), 0, 0
            nat.check(self.lib.sgr_map_step(C.byref(adam_st), stream), "sgr_map_step")
            return
        zr = self._zero
        plan = zr["plan"]
        self.comm.reduce_scatter(zr["shard"], a["flat"])
        if stored:
            # the gather pass STORED this iteration's sums for every Gaussian (grads_clean = -3) and the next one will again: the
            # 56 B x N memset per exchange is gone; whoever uses the sinks in any other way zeroes them first (_clean_flat)
            self._flat_dirty = True
        else:
            a["flat"].zero_()
            self._flat_dirty = False
        grp = (nat.SgrAdamGroup * 5)()
        r0, r1 = (C.c_int64 * 5)(), (C.c_int64 * 5)()
        for k, name in enumerate(_GROUPS):
            g = pl.groups[k]
            grp[k] = nat.SgrAdamGroup(g.param, zr["shard"].data_ptr() + 4 * plan.grad_base_offset(name), g.exp_avg, g.exp_avg_sq,
                                      g.lr, g.skip, g.step)
            r0[k], r1[k] = plan.rows[name]
        st = pl.step
        nat.check(self.lib.sgr_gaussian_adam_shard(gm._xyz.shape[0], grp, r0, r1, st.beta1, st.beta2, st.eps, float(iso_weight),
                                                   stream), "sgr_gaussian_adam_shard")
        self.comm.all_gather(zr["param"], zr["param"][plan.lo:plan.hi])
        zr["stale_moments"] = True
        nat.check(self.lib.sgr_activate(gm._xyz.shape[0], gm._scaling.data_ptr(), gm._rotation.data_ptr(), gm._opacity.data_ptr(),
                                        a["act_scale"].data_ptr(), a["act_rot"].data_ptr(), a["act_opac"].data_ptr(), stream),
                  "sgr_activate")

    def _exposure_exchange(self, rows):
        """Exposure gradients of the window rows: every camera was rendered by ONE rank, every rank steps every row."""
        if self._parallel() and self.split_views and self._exp is not None and rows:     # (own views per rank: own exposures)
            n = max(rows) + 1
            self.comm.all_reduce(self._exp.grad[:n])

    def _view(self, cam):
        vb = self._views.get(cam.uid)
        if vb is None:
            vb = _ViewBuffers(int(cam.image_height), int(cam.image_width), self.gaussians._xyz.shape[0], self.device)
            self._views[cam.uid] = vb
        if vb.gt_depth is None or vb.depth_src is not cam.depth:     # ground-truth depth stays resident on the device
            d = cam.depth
            d = torch.as_tensor(d) if not torch.is_tensor(d) else d
            vb.gt_depth = d.to(device=self.device, dtype=torch.float32).contiguous()
            vb.depth_src = cam.depth
        return vb

    def _settings(self, cam, N):
        import math
        s = nat.SgrSettings()
        s.num_gaussians, s.image_height, s.image_width = N, int(cam.image_height), int(cam.image_width)
        s.sh_degree, s.sh_coeffs = 0, 1
        s.tanfovx, s.tanfovy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        s.scale_modifier, s.prefiltered, s.debug = 1.0, 0, 0
        view, full, center = cam._matrices()
        s.bg, s.viewmatrix, s.projmatrix = self.background.data_ptr(), view.data_ptr(), full.data_ptr()
        # cameras with the same intrinsics share ONE raw projection matrix tensor: batched launches need identical
        # view-independent settings (pointer equality is what the C ABI can check)
        ik = (cam.fx, cam.fy, cam.cx, cam.cy, int(cam.image_width), int(cam.image_height))
        praw = self._proj_raw.get(ik)
        if praw is None:
            praw = self._proj_raw[ik] = cam.projection_matrix.detach().contiguous().clone()
        s.projmatrix_raw, s.campos = praw.data_ptr(), center.data_ptr()
        return s

    def _workspace(self, vb, N, H, W, cap):
        """Saved + scratch blocks of one camera.  Scratch is private per camera: the views of an iteration run as ONE
        batched launch per stage (sgr_map_views), so they cannot share it."""
        key = (N, H, W, cap)
        sz = self._ws_bytes.get(key)
        if sz is None:
            if len(self._ws_bytes) > 64:
                self._ws_bytes.clear()
            sz = self._ws_bytes[key] = (self.lib.sgr_saved_bytes(N, H, W, cap), self.lib.sgr_scratch_bytes(N, H, W, cap))
        sb, tb = sz
        # blocks are kept across map sizes and capacities (head-room: a map grows by a few per cent per keyframe).  The per-tile
        # counters sit at a fixed offset for a given (H, W), so a block that went through a forward stays clean whatever N is
        if vb.saved is None or vb.saved.numel() < sb or vb.scratch is None or vb.scratch.numel() < tb:
            self._acquire_blocks(vb, sb, tb)
        elif id(vb) in self._ws_owners:
            self._ws_owners.move_to_end(id(vb))
        vb.capacity = cap
        return nat.SgrWorkspace(vb.saved.data_ptr(), vb.saved.numel(), vb.scratch.data_ptr(), vb.scratch.numel(), cap,
                                int(vb.clean), self._max_list())

    def _acquire_blocks(self, vb, sb, tb):
        """Gives `vb` a (saved, scratch) pair of at least (sb, tb) bytes: its own if large enough, else the pair of the least
        recently used owner that the call being built does not need, else new blocks (with head-room)."""
        me = id(vb)
        self._ws_owners.pop(me, None)
        if not getattr(vb, "is_slot", False) and len(self._ws_owners) >= self.max_live_ws:
            for k, old in self._ws_owners.items():
                if k not in self._ws_protect:
                    del self._ws_owners[k]
                    if old.saved is not None and old.saved.numel() >= sb and old.scratch.numel() >= tb and (vb.saved is None):
                        vb.saved, vb.scratch, vb.clean = old.saved, old.scratch, False     # (another (H, W) may have used it)
                        vb.ran, vb.ovf_seen = False, 0      # (its header is the previous owner's until the next forward zeroes it)
                        vb.gen += 1
                    old.saved = old.scratch = None
                    old.clean = old.ran = False
                    old.ovf_seen = 0
                    old.mv = None
                    # whoever cached structs that point into the evicted blocks must not use them again (the span cache
                    # holds COPIES of the window cameras' structs and is only validated by these generations)
                    old.gen += 1
                    self._span_cache = None
                    break
        if vb.saved is None or vb.saved.numel() < sb:
            vb.saved = torch.empty((int(sb * 1.3) + (1 << 20)) // 256 * 256, dtype=torch.uint8, device=self.device)
            vb.gen += 1
            vb.clean = False       # per-tile counters of a fresh block are garbage: the library zeroes them once
            vb.ran, vb.ovf_seen = False, 0
        if vb.scratch is None or vb.scratch.numel() < tb:
            vb.scratch = torch.empty((int(tb * 1.3) + (1 << 20)) // 256 * 256, dtype=torch.uint8, device=self.device)
            vb.gen += 1
        if not getattr(vb, "is_slot", False):
            self._ws_owners[me] = vb

    def _max_list(self):
        """Longest per-tile list measured on any camera of this map (0: nothing measured yet): the library picks the sort
        build of the tile kernels from it -- a deterministic function of measurements (the capacity-derived guess it
        replaces depended on the probe history and flipped the build between two runs of the same scene)."""
        return max(self._list_hint.values()) if self._list_hint else 0

    def _read_header(self, uid, vb, own_attempts=False):
        """One synchronous header read of a camera's last forward: pair count, overflow word, longest list.  The sticky event count
        (SavedHeader.overflow_events) is acknowledged for THIS forward only -- or, `own_attempts`, for everything so far (a probe's
        own failed attempts) -- so that a truncated forward of an open transaction on the same workspace still shows at the next
        check (a forward-only render used to erase it together with the header)."""
        w = (C.c_uint32 * 16)()
        nat.check(self.lib.sgr_query_header(vb.saved.data_ptr(), w, self._stream()), "sgr_query_header")
        events = int(w[12])
        vb.ovf_seen = events if own_attempts else min(events, vb.ovf_seen + (1 if int(w[1]) else 0))
        return self._apply_header(uid, vb, w)

    def _apply_header(self, uid, vb, w):
        R, ov, longest = int(w[0]), int(w[1]), int(w[10])
        vb.pairs, vb.estimated = R, False
        if not getattr(vb, "is_slot", False):         # (a slot's count belongs to whichever camera rendered in it last)
            self._pair_hint[uid] = (R, self.gaussians._xyz.shape[0])
        if longest != self._list_hint.get(uid):
            old = self._build_class()
            self._list_hint[uid] = longest
            if self._build_class() != old:
                self._hint_changed()           # the cached SgrMapViews carry the hint
        return R, ov

    def _build_class(self):
        """What the library derives from the hint: -1 nothing measured yet, 0 a fresh map (lists <= 64: K1 keeps one view part per
        segment), 1 lists within the 256-entry buckets (no scatter launch), 2 / 3 / 4 the light / mid / heavy sort build of the tile
        kernels (sgr_common.h: k1_parts_for, kBucket; sgr_blend.hip: blend_build)."""
        m = self._max_list()
        return -1 if m == 0 else (0 if m <= 64 else (1 if m <= 256 else (2 if m <= 768 else (3 if m <= 1536 else 4))))

    def _estimate_pairs(self, cam, vb):
        """Pair count of a camera whose buffers are new.  A synchronous probe forward per camera and map size cost a converged
        session ~120 host-synchronised renders per keyframe (the 2 random views of every iteration are mostly cameras not yet
        seen at this size), so the count is carried over instead: the camera's last MEASURED count, scaled by the growth of
        the map; a camera never measured takes the largest estimate of the others (neighbouring views of one room); only
        with nothing to go by is it probed.  Capacity is twice the estimate; an estimate that still falls short is found by the
        check that closes the transaction, which re-runs the iterations at the corrected capacity (_txn_commit)."""
        est = estimate_pairs(self._pair_hint, cam.uid, self.gaussians._xyz.shape[0])
        if est is None:
            self._probe(cam, vb)
            return
        vb.pairs, vb.estimated = est, True
        if self.world == 1:        # (several ranks: the checks carry a collective and must fall on the same iterations everywhere)
            self._since_check = max(self._since_check, self.check_every - 2)      # measure soon

    def _probe(self, cam, vb):
        """One synchronous forward to learn this camera's pair count at the current map size."""
        gm = self.gaussians
        N, H, W = gm._xyz.shape[0], int(cam.image_height), int(cam.image_width)
        s = self._settings(cam, N)
        out = nat.SgrOutputs(vb.color.data_ptr(), vb.depth.data_ptr(), vb.opacity.data_ptr(), vb.radii.data_ptr(),
                             vb.n_touched.data_ptr())
        inp = self._inputs()
        a = self._acc          # the probe runs before the step's own activation: activate now (idempotent)
        nat.check(self.lib.sgr_activate(N, gm._scaling.data_ptr(), gm._rotation.data_ptr(), gm._opacity.data_ptr(),
                                        a["act_scale"].data_ptr(), a["act_rot"].data_ptr(), a["act_opac"].data_ptr(),
                                        self._stream()), "sgr_activate")
        cap, R = max(self._cap, self.capacity_floor), C.c_int64(0)
        while True:
            ws = self._workspace(vb, N, H, W, cap)
            rc = self.lib.sgr_forward(C.byref(s), C.byref(inp), C.byref(out), C.byref(ws), C.byref(R), self._stream())
            if rc == nat.SGR_ERR_CAPACITY:
                cap = int(R.value * 1.25) + 1024
                continue
            nat.check(rc, "sgr_forward")
            break
        vb.clean = vb.ran = True
        self._read_header(cam.uid, vb, own_attempts=True)            # (also the longest list: the sort build of the tile kernels)
        if vb.pairs > self.max_pairs:
            gm = self.gaussians
            with torch.no_grad():
                big = float(gm.get_scaling.max())
            raise RuntimeError(f"camera {cam.uid}: {vb.pairs} (tile, Gaussian) pairs for {N} Gaussians -- more than max_pairs = "
                               f"{self.max_pairs}; the map has degenerated (largest scale {big:.3g} m)")

    def _mark_clean(self, cams):
        """A completed forward leaves the per-tile counters of a saved block zero: later calls skip the zeroing launch."""
        for c in cams:
            vb = self._views.get(c.uid)
            if vb is None:
                continue
            vb.ran = True
            if not vb.clean:
                vb.clean = True
                for ent in (vb.mv or {}).values():
                    ent[1].ws.counters_clean = 1

    # ------------------------------------------------------------------------------------------------ pieces
    def _inputs(self):
        gm, a = self.gaussians, self._acc
        return nat.SgrInputs(gm._xyz.data_ptr(), a["act_opac"].data_ptr(), gm._features_dc.data_ptr(), None,
                             a["act_scale"].data_ptr(), a["act_rot"].data_ptr(), None)

    def _grad_sinks(self, stats):
        gm, a = self.gaussians, self._acc
        return nat.SgrGradInputs(a["xyz"].data_ptr(), None, a["opacity"].data_ptr(), a["f_dc"].data_ptr(), None,
                                 a["scaling"].data_ptr(), a["rotation"].data_ptr(), None, None, 1,
                                 gm.xyz_gradient_accum.data_ptr() if stats else None,
                                 gm.denom.data_ptr() if stats else None, gm.max_radii2D.data_ptr() if stats else None)

    def _map_view(self, cam, initialization=False, images=True, slot=False):
        """The cached SgrMapView of a camera: pointers into its persistent buffers, sized for the shared capacity.
        images=False: loss + gradients only (the rendered colour / depth / opacity are not written to HBM).
        slot=True: a pool entry of a span whose picks render in shared workspace slots: no workspace of its own."""
        vb = self._view(cam)
        # (storage addresses, not id()s: an id can be re-used by the object that replaces a freed one -- the exposure parameters are
        #  re-bound to their slab row by attach() / _grow())
        row = self._exp.row_of(cam) if self._exp is not None else None
        key = (cam._version, self._cap, cam.exposure_a.data_ptr(), cam.exposure_b.data_ptr(), cam.original_image.data_ptr(),
               vb.gt_depth.data_ptr(), initialization, self.keyframe_optimizers is not None,
               None if row is None else self._exp.grad.data_ptr() + 8 * row)
        if vb.mv is None:
            vb.mv = {}
        hit = vb.mv.get((images, slot))
        gm = self.gaussians
        N, H, W = gm._xyz.shape[0], int(cam.image_height), int(cam.image_width)
        if hit is not None and hit[0] == key:
            mv = hit[1]
            if hit[2] != N or hit[3] != vb.gen:
                # the map changed size, or one of the camera's buffers was re-allocated meanwhile (an entry that is only compared by
                # N would point at freed radii / workspace memory once N returns to an earlier value): same camera -- only what
                # depends on N or on the buffers is re-filled
                mv.settings.num_gaussians = N
                mv.out.radii, mv.out.n_touched = vb.radii.data_ptr(), vb.n_touched.data_ptr()
                if not slot:
                    mv.ws = self._workspace(vb, N, H, W, self._cap)
                hit[2], hit[3] = N, vb.gen
            elif not slot and (vb.saved is None or id(vb) not in self._ws_owners):
                mv.ws = self._workspace(vb, N, H, W, self._cap)         # (its blocks went to another camera meanwhile)
                hit[3] = vb.gen
            elif not slot:
                self._ws_owners.move_to_end(id(vb))
            return mv
        s = self._settings(cam, N)
        out = nat.SgrOutputs(vb.color.data_ptr() if images else None, vb.depth.data_ptr() if images else None,
                             vb.opacity.data_ptr() if images else None, vb.radii.data_ptr(), vb.n_touched.data_ptr())
        ws = nat.SgrWorkspace() if slot else self._workspace(vb, N, H, W, self._cap)
        mv = nat.SgrMapView()
        mv.settings, mv.out, mv.ws = s, out, ws
        mv.gt_image, mv.gt_depth = cam.original_image.data_ptr(), vb.gt_depth.data_ptr()
        mv.exposure_a = None if initialization else cam.exposure_a.data_ptr()
        mv.exposure_b = None if initialization else cam.exposure_b.data_ptr()
        mv.loss, mv.dL_dimage, mv.dL_ddepth = vb.loss.data_ptr(), vb.d_color.data_ptr(), vb.d_depth.data_ptr()
        mv.dL_dexposure = vb.d_exp.data_ptr() if row is None else self._exp.grad.data_ptr() + 8 * row
        mv.dL_dtau = vb.d_tau.data_ptr() if self.keyframe_optimizers is not None else None
        mv.loss_scratch, mv.loss_scratch_bytes = vb.loss_scratch.data_ptr(), vb.loss_scratch.numel()
        vb.mv[(images, slot)] = [key, mv, N, vb.gen]
        return mv

    def _settle_capacity(self, cams):
        """Pair-count estimates of cameras whose buffers are new, and ONE capacity for all cameras (batched launches share a
        layout).  It only grows, in steps: the workspaces and the cached structs keep their shape from keyframe to keyframe
        (check_overflow shrinks it when it is far off)."""
        need = self._cap
        for c in cams:
            vb = self._view(c)
            if vb.pairs < 0:                          # buffers are new (camera, or every camera after the map changed size)
                self._estimate_pairs(c, vb)
            need = max(need, self.capacity_floor, 2 * vb.pairs)
        if need > self._cap:
            self._cap = int(need * 1.25)
            self._views_dirty()

    def _views_array(self, cams, initialization, images=True, slot=False, settle=True):
        n = len(cams)
        if settle:
            self._settle_capacity(cams)
        if not slot:
            self._ws_protect = {id(self._view(c)) for c in cams}
            self.max_live_ws = max(self.max_live_ws, n + 8)
        return (nat.SgrMapView * n)(*[self._map_view(c, initialization, images, slot) for c in cams])

    def _span_arrays(self, window_cams, pool_cams, initialization):
        """(window array, pool array) of a span -- SgrMapView structs for sgr_map_run -- built ONCE for the spans that share them: a
        keyframe's map() call goes out as 2 + 6 + 52 iterations (plus a verified first iteration), all with the same window and the
        same pool (every other keyframe: hundreds late in a session).  Re-making both arrays and re-settling the capacity per span
        cost a 20-keyframe session 7 ms of host time per keyframe, most of it with the GPU idle behind the keyframe-selection
        read-back (scripts/session_timeline.py).  Valid while nothing they point into has changed: map size, capacity / hints
        generation, the cameras' own versions and buffers."""
        N = self.gaussians._xyz.shape[0]
        cams = list(window_cams) + list(pool_cams)
        ident = (N, bool(initialization), len(window_cams), tuple(id(c) for c in cams), tuple(c._version for c in cams))
        hit = self._span_cache if self.cache_span_arrays else None
        if hit is not None and hit[0] == ident and hit[1] == self._gen and all(self._views[c.uid].gen == g for c, g in zip(window_cams, hit[2])):
            self._ws_protect = hit[5]
            for i, c in enumerate(window_cams):          # (the array holds COPIES of the cached structs: a block that has been through a
                vb = self._views[c.uid]                  #  forward since needs no zeroing launch in front of the next span)
                if vb.clean:
                    hit[3][i].ws.counters_clean = 1
                if id(vb) in self._ws_owners:            # a hit never goes through _workspace(): keep the window's blocks the most
                    self._ws_owners.move_to_end(id(vb))  # recently used ones, or they become the likeliest eviction victims
            return hit[3], hit[4]
        self._settle_capacity(cams)                                              # estimates for new cameras, ONE capacity
        win = self._views_array(window_cams, initialization, images=False, settle=False) if window_cams else None
        protect = self._ws_protect
        pool = self._views_array(pool_cams, initialization, images=False, slot=True, settle=False) if pool_cams else None
        self._span_cache = (ident, self._gen, [self._views[c.uid].gen for c in window_cams], win, pool, protect)
        return win, pool

    def _slot_workspaces(self, count, N, H, W):
        """`count` shared workspace slots at the current capacity (SgrMapRun.pick_ws)."""
        while len(self._slots) < count:
            sl = _Slot()
            self._slots.append(sl)
        return (nat.SgrWorkspace * max(1, count))(*[self._workspace(sl, N, H, W, self._cap) for sl in self._slots[:count]])

    def _plan(self):
        """Structs that only change when the parameter tensors do (new N, opacity reset, ...)."""
        gm, a = self.gaussians, self._acc
        by_name = {g["name"]: g for g in gm.optimizer.param_groups}
        key = tuple(by_name[n]["params"][0] for n in _GROUPS) + (a["xyz"],)        # objects, compared with `is` (see _ensure_state)
        if self._plan_key is None or len(self._plan_key) != len(key) or any(x is not y for x, y in zip(self._plan_key, key)):
            pl = _Plan()
            pl.inp = self._inputs()
            pl.step = nat.SgrMapStep()
            st = pl.step
            st.num_gaussians = gm._xyz.shape[0]
            st.scaling, st.rotation, st.opacity = gm._scaling.data_ptr(), gm._rotation.data_ptr(), gm._opacity.data_ptr()
            st.scales_out, st.rot_out, st.opac_out = a["act_scale"].data_ptr(), a["act_rot"].data_ptr(), a["act_opac"].data_ptr()
            st.in_ = C.pointer(pl.inp)
            tr = self.config["mapping"]["Training"]
            st.alpha, st.rgb_boundary_threshold = float(tr.get("alpha", 0.95)), float(tr["rgb_boundary_threshold"])
            pl.groups = (nat.SgrAdamGroup * 5)()
            pl.states = []
            for k, name in enumerate(_GROUPS):
                g = by_name[name]
                p = g["params"][0]
                stt = gm.optimizer.state[p]
                pl.states.append((g, stt))
                pl.groups[k] = nat.SgrAdamGroup(p.data_ptr(), a[name].data_ptr(), stt["exp_avg"].data_ptr(),
                                                stt["exp_avg_sq"].data_ptr(), float(g["lr"]), 0, int(stt["step"].item()))
            pl.frest_state = gm.optimizer.state[by_name["f_rest"]["params"][0]]
            st.beta1, st.beta2 = gm.optimizer.param_groups[0]["betas"]
            st.eps = gm.optimizer.param_groups[0]["eps"]
            pl.gi_stats = self._grad_sinks(True)
            pl.gi_nostats = self._grad_sinks(False)
            self._plan_obj, self._plan_key = pl, key
        return self._plan_obj

    def _setup(self, pl, iso_weight, adam, skip, stats, forward_only, exposure, bump=True, lrs=None):
        """Fills the view-independent part of pl.step.  bump: advance the Adam step counters here (one iteration);
        sgr_map_run advances them itself.  lrs: learning rates by group name (default: the optimiser's current ones)."""
        st = pl.step
        st.forward_only = int(forward_only)
        st.grads_clean = int(self._acc_clean) if self.fuse_tail else -1
        st.grads = C.pointer(pl.gi_stats if stats else pl.gi_nostats)
        if adam and not forward_only:
            for k, (g, stt) in enumerate(pl.states):
                grp = pl.groups[k]
                grp.lr = float(g["lr"]) if lrs is None else lrs[_GROUPS[k]]
                if _GROUPS[k] in skip:
                    grp.skip = 1
                else:
                    grp.skip = 0
                    if bump:
                        grp.step += 1
                        stt["step"] += 1
            if bump:
                pl.frest_state["step"] += 1
            st.adam_groups = pl.groups
            st.iso_weight = float(iso_weight)
        else:
            st.adam_groups = None
        st.exp_rows = 0
        if exposure != "none" and self._exp is not None and self._exp_rows and not forward_only:
            e = self._exp
            if exposure == "window":
                first, n, active = 0, max(self._exp_rows) + 1, e.active
            elif exposure == "per_pick":            # sgr_map_run offsets the row itself (pool_exp_row)
                first, n, active = 0, 1, e.ones
            else:                                   # only the rows in `exposure` (final_refine: the rendered camera)
                rows = [r for r in exposure if r in self._exp_rows]
                first, n, active = (rows[0], 1, e.ones) if rows else (0, 0, e.ones)
            if n:
                st.exp_rows, st.exp_row_width = n, 2
                st.exp_param, st.exp_grad = e.param.data_ptr() + 8 * first, e.grad.data_ptr() + 8 * first
                st.exp_avg, st.exp_avg_sq = e.m.data_ptr() + 8 * first, e.v.data_ptr() + 8 * first
                st.exp_step, st.exp_active = e.step.data_ptr() + 4 * first, active.data_ptr()
                st.exp_lr, st.exp_beta1, st.exp_beta2, st.exp_eps = 0.01, 0.9, 0.999, 1e-8
        return st

    def _run_span(self, window_cams, pool_cams, picks, lrs, iso_weight, exposure, stats=True, initialization=False, verified=False):
        """len(lrs) regular iterations with ONE host call (sgr_map_run): iteration k renders window_cams plus
        pool_cams[picks[k]] and steps Adam with the xyz learning rate lrs[k].
        A view whose workspace is sized by a carried-over ESTIMATE of its pair count (buffers are new after the map changed
        size) could be truncated for the whole span without anybody looking: the first iteration then runs on its own and
        the headers of its views are read back before the rest of the span is enqueued (one synchronisation per span that
        follows a change of the map; views that only appear later in the span are covered by the periodic check, which warns)."""
        n_it = len(lrs)
        if self._parallel() or n_it == 0:
            raise RuntimeError("_run_span is the single-GPU fast path")
        self._span_arrays(window_cams, pool_cams, initialization)                # estimates for new cameras, ONE capacity (cached per map size)
        per0 = len(picks) // n_it if picks else 0
        if (n_it > 1 and not verified and not self._replaying and self.verify_estimates
                and any(self._views[c.uid].estimated for c in list(window_cams) + [pool_cams[k] for k in picks[:per0]])):
            self._run_span(window_cams, pool_cams, picks[:per0], lrs[:1], iso_weight, exposure, stats, initialization, verified=True)
            # the launch structs of the rest are built while that iteration runs; the read-back then only decides whether they stand
            rest = (window_cams, pool_cams, picks[per0:], lrs[1:], iso_weight, exposure, stats, initialization)
            prep = self._prepare_span(*rest)
            if not self._txn_commit():               # (an estimate that was short: the iteration has been re-run at the right capacity)
                for k in picks[:per0]:               # (rendered in the slots that were just read back: their estimates held)
                    self._views[pool_cams[k].uid].estimated = False
            return self._span_entry(rest, prep)
        return self._span_entry((window_cams, pool_cams, picks, lrs, iso_weight, exposure, stats, initialization))

    def _span_entry(self, args, prep=None):
        """Enqueues a span as part of the open transaction.  `prep`: launch structs built ahead of time -- used if the capacity /
        sort build they were made for still stands (a replay after a correction builds its own)."""
        args = tuple(list(a) if isinstance(a, (list, tuple)) else a for a in args)
        box = [prep]

        def fn():
            pr, box[0] = box[0], None
            if pr is None or pr[-1] != self._gen:
                pr = self._prepare_span(*args)
            self._launch_span(pr)
        return self._txn_do(fn)

    def _prepare_span(self, window_cams, pool_cams, picks, lrs, iso_weight, exposure, stats, initialization):
        """Everything sgr_map_run needs, as ctypes objects (kept alive by the returned tuple)."""
        n_it = len(lrs)
        pl = self._plan()
        win, pool = self._span_arrays(window_cams, pool_cams, initialization)    # (ONE capacity for the window and the pool)
        st = self._setup(pl, iso_weight, True, (), stats, False, exposure, bump=False)
        per = len(picks) // n_it if picks else 0
        run = nat.SgrMapRun()
        run.step = st
        run.num_iters, run.num_window, run.pool_size, run.picks_per_iter = n_it, len(window_cams), len(pool_cams), per
        if win is not None:
            run.window = win
        if pool is not None:
            run.pool = pool
        pk = (C.c_int32 * max(1, len(picks)))(*picks)
        lr = (C.c_float * n_it)(*lrs)
        run.picks, run.lr0, run.adam_groups = pk, lr, pl.groups
        run.n_touched_last_only = 1
        slots = None
        if pool_cams and per:
            c0 = pool_cams[0]
            slots = self._slot_workspaces(per, self.gaussians._xyz.shape[0], int(c0.image_height), int(c0.image_width))
            run.pick_ws = slots
        rows = None
        if exposure == "per_pick":
            rows = (C.c_int32 * max(1, len(pool_cams)))(*[
                (self._exp.row_of(c) if self._exp is not None and self._exp.row_of(c) in self._exp_rows else -1)
                for c in pool_cams])
            run.pool_exp_row = rows
        return run, pl, n_it, per, list(window_cams), list(pool_cams), list(picks), (win, pool, pk, lr, slots, rows), self._gen

    def _launch_span(self, prep):
        run, pl, n_it, per, window_cams, pool_cams, picks, _keep, _gen = prep
        self._clean_flat()
        rc = self.lib.sgr_map_run(C.byref(run), self._stream())
        nat.check(rc, "sgr_map_run")
        self.gaussians.invalidate_activations()    # parameters changed through raw pointers: cached torch activations are stale
        self._acc_clean = True
        self._mark_clean(window_cams)
        for sl in self._slots[:per]:
            sl.clean = sl.ran = True
        self._last_pick_slots = {pool_cams[k].uid: j for j, k in enumerate(picks[len(picks) - per:])} if per else {}
        for g, stt in pl.states:                 # the library advanced pl.groups[k].step; mirror it in torch's state
            stt["step"] += n_it
        pl.frest_state["step"] += n_it

    def _run_span_ranks(self, window_cams, pool_cams, picks, lrs, iso_weight, exposure, stats=True):
        args = (list(window_cams), list(pool_cams), list(picks), list(lrs), iso_weight, exposure, stats)
        return self._txn_do(lambda: self._run_span_ranks_impl(*args))

    def _run_span_ranks_impl(self, window_cams, pool_cams, picks, lrs, iso_weight, exposure, stats=True):
        """The multi-GPU form of _run_span.  Per iteration: (1) this rank's share of the views, gradient sums added into the
        flat buffer (the gather pass of the fused form, no optimiser; the loss sums ride in that launch); (2) the exchange:
        reduce-scatter -> Adam on this rank's rows -> all-gather (or all-reduce -> replicated Adam); (3) the exposure rows,
        all-reduced (a few floats) and stepped identically everywhere.  The collectives sit between two kernels of the same
        iteration, so the iterations are enqueued from a lean host loop instead of one sgr_map_run."""
        n_it = len(lrs)
        pl = self._plan()
        nw, per = len(window_cams), (len(picks) // n_it if picks else 0)
        mine_w = self._local(list(window_cams) + [None] * per)               # positions of an iteration this rank renders
        pos = list(range(nw + per))[self.rank::self.world] if self.split_views else list(range(nw + per))
        del mine_w
        cams_needed = [c for i, c in enumerate(window_cams) if i in pos] + (list(pool_cams) if any(i >= nw for i in pos) else [])
        if cams_needed:
            self._settle_capacity(cams_needed)                             # estimates for new cameras, ONE capacity
        st = self._setup(pl, iso_weight, True, (), stats, False, "none", bump=False)
        views_st = nat.SgrMapStep()
        C.memmove(C.byref(views_st), C.byref(st), C.sizeof(st))
        # store mode (-3) while the sinks hold nothing that must survive: the gather pass overwrites every row, nobody zeroes the
        # flat buffer between iterations; a prune pass's leftovers (not _acc_clean) are ADDED to (-2) and the buffer zeroed after
        stored = self._acc_clean and self._zero is not None and self.keyframe_optimizers is None      # (pose gradients: no fused gather pass, see _step)
        if not stored:
            self._clean_flat()
        views_st.adam_groups, views_st.grads_clean, views_st.exp_rows = None, (-3 if stored else -2), 0
        arr = (nat.SgrMapView * max(1, len(pos)))()
        views_st.num_views, views_st.views = len(pos), arr
        stream = self._stream()
        gm, a = self.gaussians, self._acc
        nat.check(self.lib.sgr_activate(gm._xyz.shape[0], gm._scaling.data_ptr(), gm._rotation.data_ptr(), gm._opacity.data_ptr(),
                                        a["act_scale"].data_ptr(), a["act_rot"].data_ptr(), a["act_opac"].data_ptr(), stream),
                  "sgr_activate")
        views_st.scaling, views_st.rotation, views_st.opacity = None, None, None      # (every exchange ends with fresh activations)
        exp_rows = list(self._exp_rows) if (exposure != "none" and self._exp is not None) else []
        used = []
        for k in range(n_it):
            used = list(window_cams) + [pool_cams[picks[k * per + j]] for j in range(per)]
            self._ws_protect = {id(self._view(used[i])) for i in pos}      # (workspaces are handed out lazily, least recently used first)
            for slot, i in enumerate(pos):
                arr[slot] = self._map_view(used[i], False, images=False)
            if exp_rows:
                self._exp.grad[: max(exp_rows) + 1].zero_()                 # rows of cameras other ranks render stay 0 here
            if pos:
                nat.check(self.lib.sgr_map_step(C.byref(views_st), stream), "sgr_map_step")
            elif stored:
                self._clean_flat()           # (a rank without views this iteration contributes zeros)
            pl.groups[0].lr = lrs[k]
            for g in range(5):
                pl.groups[g].step += 1
            self._exchange_and_adam(pl, iso_weight, stored=stored and bool(pos))
            if exp_rows:
                self._exposure_exchange(exp_rows)
                self._exp.step_mask(self.lib, exp_rows, stream)
        for g, stt in pl.states:
            stt["step"] += n_it
        pl.frest_state["step"] += n_it
        self.gaussians.invalidate_activations()
        self._acc_clean = True
        self._mark_clean([used[i] for i in pos])

    def _step(self, cams, iso_weight=0.0, adam=True, skip=(), initialization=False, stats=True, forward_only=False,
              exposure="none", activate=True):
        """ONE host call: activate -> batched views (forward / loss / backward) -> Adam -> exposure Adam.  Anything but a
        forward-only render is part of the open transaction (see _txn_commit); the learning rates it steps with are the ones of
        THIS moment, also when it is issued again."""
        if forward_only:
            return self._step_impl(cams, iso_weight, adam, skip, initialization, stats, True, exposure, activate, None)
        cams = list(cams)
        exposure = list(exposure) if isinstance(exposure, list) else exposure
        lrs = {g["name"]: float(g["lr"]) for g in self.gaussians.optimizer.param_groups}
        return self._txn_do(lambda: self._step_impl(cams, iso_weight, adam, tuple(skip), initialization, stats, False, exposure, activate, lrs))

    def _step_impl(self, cams, iso_weight, adam, skip, initialization, stats, forward_only, exposure, activate, lrs):
        pl = self._plan()
        par = self._parallel() and not forward_only
        all_cams = cams
        if par:
            cams = self._local(cams)
            if self._exp is not None and self._exp_rows and len(all_cams):
                self._exp.grad[: max(self._exp_rows) + 1].zero_()        # rows of cameras other ranks render stay 0 here
        arr = self._views_array(cams, initialization)
        if adam and not forward_only and "scaling" not in skip and self._stale_iso:
            iso_weight, self._stale_iso = iso_weight + self._stale_iso, 0.0     # the prune pass's share rides in this step
        st = self._setup(pl, iso_weight, adam, skip, stats, forward_only, exposure, lrs=lrs)
        st.num_views, st.views = len(cams), arr
        if not activate:
            sc, st.scaling = st.scaling, None
            ro, st.rotation = st.rotation, None
            op, st.opacity = st.opacity, None
        if par:
            # multi-GPU: (1) this rank's views into the flat sinks, (2) the exchange + the identical Adam step everywhere
            # (see _run_span_ranks); a call without an optimiser step just leaves the local sums in the sinks
            do_adam = bool(st.adam_groups)
            iso = st.iso_weight if do_adam else 0.0
            st.adam_groups, st.exp_rows = None, 0
            # store mode (see _run_span_ranks): this rank's views overwrite every row of the sinks, which hold nothing to keep
            # (only the fused gather pass can store, and a batch that asks for pose gradients -- mapping.BA -- does not take it: the
            #  library then zeroes the sinks itself before its accumulating passes (sgr_api.hip: store_sinks), but there is nothing to
            #  gain, so such iterations keep the add mode)
            stored = (do_adam and self.fuse_tail and self._acc_clean and self._zero is not None and len(cams) > 0
                      and self.keyframe_optimizers is None)
            if not stored:
                self._clean_flat()
            if do_adam and self.fuse_tail:
                st.grads_clean = -3 if stored else -2
            rc = self.lib.sgr_map_step(C.byref(st), self._stream()) if (len(cams) or activate) else 0
            if rc == 0 and do_adam:
                self._exchange_and_adam(pl, iso, skip, stored=stored)
                if exposure != "none" and len(all_cams):     # (never with pose optimisation: map() then passes "none" and steps after)
                    self._exposure_slab_step(all_cams, only_rendered=isinstance(exposure, list))
        else:
            if not forward_only:
                self._clean_flat()
            rc = self.lib.sgr_map_step(C.byref(st), self._stream())
        if not activate:
            st.scaling, st.rotation, st.opacity = sc, ro, op
        nat.check(rc, "sgr_map_step")
        self._mark_clean(cams)
        if not forward_only:
            if adam:
                self.gaussians.invalidate_activations()
                self._acc_clean = True
            elif len(cams):
                self._acc_clean = False

    def _run_views(self, cams, initialization=False, stats=True, forward_only=False):
        """Views only (no activation, no optimiser step): used by tests and forward-only passes."""
        self._step(cams, adam=False, initialization=initialization, stats=stats, forward_only=forward_only, activate=False)

    def _view_step(self, cam, initialization=False, stats=True):
        self._run_views([cam], initialization=initialization, stats=stats)
        return self._views[cam.uid]

    def _activate(self):
        self._step([], adam=False)

    def _forward(self, cam, vb=None):
        self._run_views([cam], forward_only=True)

    def _adam(self, iso_weight, skip=()):
        self._step([], iso_weight=iso_weight, adam=True, skip=skip, activate=False)

    def render_forward(self, viewpoint):
        """Forward-only render used for keyframe selection (mapper.py:972-978): buffers of the camera, no autograd."""
        if self.autograd_fallback:
            return MappingLoop.render_forward(self, viewpoint)
        self._ensure_state()
        self._step([viewpoint], adam=False, forward_only=True)
        vb = self._views[viewpoint.uid]
        if viewpoint.uid not in self._pair_hint or self._pair_hint[viewpoint.uid][1] != self.gaussians._xyz.shape[0]:
            # the caller reads the images right away (keyframe selection): one read-back more measures this camera's pair count,
            # so that the mapping iterations that follow size their workspaces from a measurement, not an estimate
            R, ov = self._read_header(viewpoint.uid, vb)
            if ov:                                    # the estimate was short: grow and render again
                self._cap = max(self._cap, 2 * vb.pairs)
                self._views_dirty()
                self.overflow_events += 1
                self._step([viewpoint], adam=False, forward_only=True)
                vb = self._views[viewpoint.uid]
        return {"render": vb.color, "depth": vb.depth, "opacity": vb.opacity, "radii": vb.radii, "n_touched": vb.n_touched,
                "visibility_filter": vb.radii > 0}

    # ---- exposure (keyframe) optimiser: slab-resident parameters + one masked Adam launch
    def build_keyframe_optimizers(self):
        """mapper.py:1067-1111.  Exposure parameters live in a device slab (Camera.exposure_a/b are views of their
        row); a fresh optimiser per keyframe = moments and step counters of the window rows reset."""
        if self.autograd_fallback:
            return MappingLoop.build_keyframe_optimizers(self)
        if self._exp is None:
            self._exp = _ExposureSlab(self.device)
        rows = []
        for cam_idx in range(len(self.current_window)):
            if self.current_window[cam_idx] == 0:
                continue
            cam = self.viewpoints[self.current_window[cam_idx]]
            rows.append(self._exp.attach(cam))
        for cam in self.viewpoints.values():
            self._exp.attach(cam)                      # every camera writes its exposure gradient into its slab row
        self._exp.reset(rows)
        self._exp_rows = rows
        # (no _views_dirty() here any more: what a cached launch struct takes from the exposure slab -- the addresses of the camera's
        #  a / b and of its gradient row -- is part of the struct's key (_map_view).  Dropping every camera's structs at EVERY keyframe
        #  cost a late session ~190 rebuilds = 3-5 ms per keyframe with the GPU idle, growing with the number of keyframes.)
        self._span_cache = None
        if not self.keep_structs_over_keyframes:
            self._views_dirty()
        pose_opt = bool(self.config["mapping"]["BA"]) and not self.config["mapping"]["Training"].get("gt_camera", False)
        self.keyframe_optimizers = None
        if pose_opt:                                   # pose deltas (off by default) stay on torch.optim.Adam
            lr = self.config["mapping"]["Training"]["lr"]
            frames_to_optimize = self.config["mapping"]["Training"]["pose_window"]
            groups = []
            for cam_idx in range(min(frames_to_optimize, len(self.current_window))):
                if self.current_window[cam_idx] == 0:
                    continue
                cam = self.viewpoints[self.current_window[cam_idx]]
                groups.append({"params": [cam.cam_rot_delta], "lr": lr["cam_rot_delta"] * 0.5})
                groups.append({"params": [cam.cam_trans_delta], "lr": lr["cam_trans_delta"] * 0.5})
            self.keyframe_optimizers = torch.optim.Adam(groups) if groups else None

    def _hint_changed(self):
        """The longest-list hint moved into another build class: the cached launch structs are PATCHED (one field), not dropped.
        Dropping them (rounds 3-5: _views_dirty) re-made every camera's settings struct -- and late in a session, where the longest
        list hovers around a class boundary, that happened at almost every keyframe: ~190 struct rebuilds per keyframe at 150
        keyframes, with the GPU idle behind the capacity check (scripts/profile_session.py)."""
        hint = self._max_list()
        for vb in self._views.values():
            for ent in (vb.mv or {}).values():
                ent[1].ws.max_list_hint = hint
        self._span_cache = None            # (its arrays hold COPIES of the structs)

    def _views_dirty(self):
        self._gen += 1
        for vb in self._views.values():
            vb.mv = None
        for sl in self._slots:          # (their headers describe forwards at the old capacity / hint: not to be read again)
            sl.ran = False

    def _exposure_step(self, cams, only_rendered=False):
        """The keyframe optimisers' step after an iteration that did not carry it in its own launch: exposure rows (slab, part of
        the open transaction) and -- with mapping.BA -- the pose deltas (torch Adam; the transaction is closed first, so the pose
        gradients it consumes come from forwards that are known to be complete)."""
        cams = list(cams)
        self._txn_do(lambda: self._exposure_slab_step(cams, only_rendered))
        if self.keyframe_optimizers is not None:
            self._txn_commit()
            self._pose_step(cams)

    def _exposure_slab_step(self, cams, only_rendered=False):
        if self._exp is None or not self._exp_rows:
            pass
        elif only_rendered:       # final_refine: torch's Adam only touches parameters that HAVE a gradient: the rendered
            rows = {self._exp.row_of(cam) for cam in cams}        # camera and whatever a prune pass left a .grad on
            rendered = [r for r in rows if r in self._exp_rows]
            stale_only = [r for r in self._exp_rows if r in self._exp.stale_rows and r not in rows]
            if stale_only:
                self._exp.grad[_index(stale_only, self.device)] = 0
            self._exp.add_stale(self._exp_rows)
            self._exposure_exchange(self._exp_rows)
            for row in sorted(set(rendered) | set(stale_only)):
                self._exp.step_rows(self.lib, row, 1, self._stream())
        else:
            self._exp.add_stale(self._exp_rows)
            self._exposure_exchange(self._exp_rows)
            self._exp.step_mask(self.lib, self._exp_rows, self._stream())

    def _pose_step(self, cams):
        if self.keyframe_optimizers is not None:
            if self._parallel() and self.split_views:
                # every camera of the iteration was rendered by ONE rank (view i by rank i mod world): the pose gradients meet in one
                # small all-reduce, every rank then takes the identical pose step (replicas stay bitwise equal)
                tau = torch.zeros((len(cams), 6), dtype=torch.float32, device=self.device)
                for i, cam in enumerate(cams):
                    if i % self.world == self.rank:
                        tau[i] = self._views[cam.uid].d_tau
                self.comm.all_reduce(tau)
                taus = [tau[i] for i in range(len(cams))]
            else:
                taus = [self._views[cam.uid].d_tau for cam in cams]
            for cam, t in zip(cams, taus):
                if cam.cam_rot_delta.requires_grad:
                    cam.cam_trans_delta.grad = t[:3].clone()
                    cam.cam_rot_delta.grad = t[3:].clone()
            self.keyframe_optimizers.step()
            self.keyframe_optimizers.zero_grad(set_to_none=True)

    def _check_targets(self):
        todo = [(uid, vb) for uid, vb in self._views.items() if vb.saved is not None and vb.mv and vb.ran]
        todo += [(("slot", j), sl) for j, sl in enumerate(self._slots) if sl.saved is not None and sl.ran]
        return todo                                   # (no forward at this map size yet: the header would be an old one)

    def _post_headers(self, todo):
        """The 64-byte headers of the workspaces in `todo`, gathered on the device and copied to pinned host memory with ONE
        transfer (a read-back per camera cost a converged session ~30 host round trips per check with the GPU idle)."""
        n = len(todo)
        if self._hdr_pinned is None or self._hdr_pinned.numel() < 64 * n:
            self._hdr_pinned = torch.empty(64 * max(64, 2 * n), dtype=torch.uint8, pin_memory=True)
        host = self._hdr_pinned[: 64 * n]
        host.copy_(torch.cat([vb.saved[:64] for _, vb in todo]), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return host, ev

    def _apply_headers(self, todo, host):
        """What a capacity check does with the headers it read: pair counts and list hints are refreshed, the capacity follows the
        largest count, and the cameras whose forward was truncated are returned (the caller replays the transaction)."""
        words = host.numpy().view(np.uint32).reshape(-1, 16)
        worst, overflowed = 0, []
        for (uid, vb), w in zip(todo, words):
            # `overflow` (word 1) describes the workspace's LAST forward; words 12 / 13 are sticky (K2): the number of truncated
            # forwards since the block was zeroed and the largest pair count any forward demanded.  A span runs ~52 forwards per
            # workspace between two checks and a slot renders a different camera each time: the verdict is the CHANGE of the count.
            events = int(w[12])
            ov = int(w[1])
            if events != vb.ovf_seen and ov == 0:
                ov = 1
            vb.ovf_seen = events
            self._apply_header(uid, vb, w)
            if ov == 1:
                vb.pairs = max(vb.pairs, int(w[13]))          # size the replay by the worst forward, not by the last one
            if vb.pairs > self.max_pairs:
                raise RuntimeError(f"camera {uid}: {vb.pairs} (tile, Gaussian) pairs -- more than max_pairs = {self.max_pairs}; "
                                   "the map has degenerated")
            worst = max(worst, vb.pairs)
            if ov == 2:
                raise RuntimeError(f"camera {uid}: more than 65280 splats on one 8x8 tile -- the map has degenerated")
            if ov:
                self.overflow_events += 1
                overflowed.append(uid)
        if worst * 1.5 > self._cap or (worst > 0 and 8 * worst < self._cap and self._cap > self.capacity_floor):
            self._cap = max(self.capacity_floor, int(2.5 * worst))
            self._views_dirty()
        elif overflowed:                               # (cannot happen: an overflowed header carries a count beyond the capacity)
            self._cap = int(1.5 * self._cap)
            self._views_dirty()
        return overflowed

    def _read_overflows(self):
        """One synchronisation: the headers of every workspace that ran a forward at this map size.  Returns the cameras (or
        slots) whose last forward was truncated; the capacity has been corrected."""
        todo = self._check_targets()
        if not todo:
            return []
        host, ev = self._post_headers(todo)
        ev.synchronize()
        return self._apply_headers(todo, host)

    # ---- transactions: never drop a view (see __init__) -----------------------------------------------------------------------
    def _txn_tensors(self):
        """Every device tensor an iteration of any of the loops mutates in place (parameters, Adam moments, densification
        statistics, next-iteration activations, the exposure slab, the gradient sinks when they hold something)."""
        gm = self.gaussians
        ts = []
        for g in gm.optimizer.param_groups:
            p = g["params"][0]
            if p.numel() == 0:
                continue
            ts.append(p.data)
            st = gm.optimizer.state.get(p)
            if st:
                ts += [st["exp_avg"], st["exp_avg_sq"]]
        ts += [gm.xyz_gradient_accum, gm.denom, gm.max_radii2D]
        a = self._acc
        if a is not None:
            ts += [a["act_scale"], a["act_rot"], a["act_opac"]]
        if self._exp is not None:
            e = self._exp
            ts += [e.param, e.grad, e.m, e.v, e.step, e.stale]
        return ts

    def _snapshot_buffers(self, ts):
        """One buffer per tensor of `ts`: views of ONE byte store that is kept with head-room.  The map changes size at every keyframe
        (new Gaussians, prune passes), and eighteen torch.empty_like of never-seen sizes were eighteen allocator misses -- 1.5 ms per
        snapshot, two or three snapshots per keyframe with the GPU idle behind them (scripts/micro/txn_begin_parts.py: 4 ms per
        keyframe of a young session)."""
        if not self.snapshot_store:
            return [torch.empty_like(t) for t in ts]
        offs, need = [], 0
        for t in ts:
            offs.append(need)
            need += (t.numel() * t.element_size() + 255) // 256 * 256
        if self._txn_store is None or self._txn_store.numel() < need:
            self._txn_store = self._txn_pool = None          # (freed first: the allocator may hand the block back for the larger one)
            self._txn_store = torch.empty(int(need * 1.5) + (1 << 20), dtype=torch.uint8, device=self.device)
        return [self._txn_store[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape) for o, t in zip(offs, ts)]

    def _txn_begin(self):
        """Opens a transaction in front of the first state-changing launch after a check (no-op while one is open)."""
        if self._txn is not None or self._replaying:
            return
        gm = self.gaussians
        ts = self._txn_tensors()
        # the sinks are all-zero between optimiser steps unless a prune pass / a store-mode exchange left something in them: only
        # then are they copied (56 B per Gaussian); otherwise a restore zeroes them
        flat_live = self._acc is not None and ((not self._acc_clean) or self._flat_dirty)
        if flat_live:
            ts.append(self._acc["flat"])
        pool = self._txn_pool
        if pool is None or len(pool) != len(ts) or any(b.shape != t.shape or b.dtype != t.dtype for b, t in zip(pool, ts)):
            pool = self._txn_pool = self._snapshot_buffers(ts)
        with torch.no_grad():
            for dt in {t.dtype for t in ts}:
                torch._foreach_copy_([b for b, t in zip(pool, ts) if t.dtype == dt], [t for t in ts if t.dtype == dt])
        txn = _Plan()
        txn.tensors, txn.saved, txn.flat_live, txn.journal = ts, pool, flat_live, []
        steps = [gm.optimizer.state[g["params"][0]]["step"] for g in gm.optimizer.param_groups
                 if gm.optimizer.state.get(g["params"][0])]
        txn.py = {"steps": [(t, float(t)) for t in steps],
                  "stale_iso": self._stale_iso, "acc_clean": self._acc_clean, "flat_dirty": self._flat_dirty,
                  "exp_stale_rows": set(self._exp.stale_rows) if self._exp is not None else None,
                  "stale_moments": self._zero.get("stale_moments") if self._zero is not None else None}
        self._txn = txn

    def _txn_do(self, fn):
        """Runs a state-changing piece of enqueue work as part of the open transaction (opens one if needed)."""
        if self._replaying:
            return fn()
        self._txn_begin()
        self._txn.journal.append(fn)
        return fn()

    def _txn_restore(self):
        txn = self._txn
        with torch.no_grad():
            for dt in {t.dtype for t in txn.tensors}:
                torch._foreach_copy_([t for t in txn.tensors if t.dtype == dt], [b for b, t in zip(txn.saved, txn.tensors) if t.dtype == dt])
            if not txn.flat_live and self._acc is not None:
                self._acc["flat"].zero_()
        py = txn.py
        for t, v in py["steps"]:
            t.fill_(v)
        # the launch structs' own step counters (sgr_map_run advances them in place) follow torch's restored ones -- whichever plan is
        # current: the plan of a new parameter set is only made INSIDE the transaction's first launch (after the snapshot), so a
        # comparison with the plan seen at the snapshot would leave its counters advanced (found by a session-level run: a replay
        # right after a keyframe's new Gaussians was off by a span's worth of Adam bias correction)
        pl = self._plan_obj if self._plan_key is not None else None
        if pl is not None:
            for k, (g, stt) in enumerate(pl.states):
                pl.groups[k].step = int(float(stt["step"]))
        self._stale_iso, self._acc_clean, self._flat_dirty = py["stale_iso"], py["acc_clean"], py["flat_dirty"]
        if self._exp is not None and py["exp_stale_rows"] is not None:
            self._exp.stale_rows = set(py["exp_stale_rows"])
        if self._zero is not None:
            self._zero["stale_moments"] = py["stale_moments"]
        self.gaussians.invalidate_activations()

    def _txn_commit(self):
        """The capacity check that closes a transaction.  Returns the cameras whose forward was truncated at the FIRST look
        ([] in the regular case).  If there are any -- on any rank -- the state is put back to where the transaction began and its
        launches are issued again at the corrected capacity, until a check comes back clean: the optimisation ends bit for bit
        where a run with ample capacity ends (tests/test_gpu_round5.py)."""
        self._since_check = 0
        first = None
        for attempt in range(6):
            overflowed = self._read_overflows()
            if first is None:
                first = overflowed
            again = bool(overflowed)
            if self.world > 1 and self.comm is not None:       # the ranks replay together (their collectives are part of the journal)
                import torch.distributed as dist
                flag = torch.tensor([1.0 if again else 0.0], dtype=torch.float32, device=self.device)
                self.comm.all_reduce(flag, op=dist.ReduceOp.MAX)
                again = bool(flag.item() > 0)
            if not again or self._txn is None or not self._txn.journal:
                break
            if attempt == 5:
                raise RuntimeError("FusedMappingLoop: the pair capacity still overflows after five corrections")
            self.replayed_transactions += 1
            self._txn_restore()
            self._replaying = True
            try:
                for fn in self._txn.journal:
                    fn()
            finally:
                self._replaying = False
        self._txn = None
        return first or []

    def check_overflow(self, wait=True):
        """Did any camera's forward exceed the pair capacity since the last check?  One synchronisation; the answer covers
        everything enqueued so far, and whatever was affected has been re-run (see _txn_commit) when this returns."""
        return self._txn_commit()

    def _periodic_check(self, steady=False):
        self._txn_commit()

    def _tick(self):
        self._since_check += 1
        if self._since_check >= self.check_every:
            self._txn_commit()

    # ------------------------------------------------------------------------------------------------ loops
    def initialize_map(self, cur_frame_idx, viewpoint, iters=None):
        """mapper.py:303-398.  One view per iteration: with several ranks every rank runs the identical iterations
        (deterministic kernels keep the replicas bit-identical; there is nothing to exchange)."""
        if self.autograd_fallback:
            return MappingLoop.initialize_map(self, cur_frame_idx, viewpoint, iters)
        self._sync_moments()
        self._replicated += 1
        try:
            return self._initialize_map(cur_frame_idx, viewpoint, iters)
        finally:
            self._replicated -= 1

    def _initialize_map(self, cur_frame_idx, viewpoint, iters=None):
        vb, nt = None, None
        total = self.init_itr_num if iters is None else iters
        mapping_iteration = 0
        while mapping_iteration < total:
            # regular iterations up to the next densify / reset point: ONE host call (the learning rate is not scheduled
            # during initialisation, mapper.py:303-353)
            n = 0
            # (the last iteration goes through _step: its rendered images are the return value, mapper.py:355-398)
            while (self.span_calls and mapping_iteration + n < total - 1
                   and (mapping_iteration + n) % self.init_gaussian_update != 0
                   and self.iteration_count + n + 1 != self.init_gaussian_reset
                   and self.iteration_count + n + 1 != self.opt_params.densify_from_iter):
                n += 1
            if n > 0:
                self._ensure_state()
                lr = float(self._xyz_group()["lr"])
                self._run_span([viewpoint], [], [], [lr] * n, 0.0, "none", initialization=True)
                self.iteration_count += n
                mapping_iteration += n
                vb = self._views[viewpoint.uid]
                nt = vb.n_touched
                self._since_check += n
                if self._since_check >= self.check_every:
                    self._txn_commit()
                continue
            self.iteration_count += 1
            self._ensure_state()
            densify = mapping_iteration % self.init_gaussian_update == 0
            reset = self.iteration_count == self.init_gaussian_reset or (
                self.iteration_count == self.opt_params.densify_from_iter)
            mapping_iteration += 1
            if densify or reset:
                # the reference densifies / resets between backward and optimizer.step (mapper.py:339-352): tensors
                # re-created there have grad None and are skipped by Adam
                self._step([viewpoint], adam=False, initialization=True)
                self._txn_commit()            # the surgery below consumes statistics and gradients: verified complete first
                vb = self._views[viewpoint.uid]
                nt = vb.n_touched
                with torch.no_grad():
                    if densify:
                        self.gaussians.densify_and_prune(self.opt_params.densify_grad_threshold, self.init_gaussian_th,
                                                         self.init_gaussian_extent, None)
                    if reset:
                        self.gaussians.reset_opacity()
                if densify:
                    self._acc_key = None             # every tensor was re-created: no step at all this iteration
                    continue
                self._step([], adam=True, skip=("opacity",), activate=False)
            else:
                self._step([viewpoint], adam=True, initialization=True)
                vb = self._views[viewpoint.uid]
                nt = vb.n_touched
            self._tick()
        self._txn_commit()
        # like the reference, visibility comes from the LAST iteration's render (mapper.py:355)
        self.occ_aware_visibility[cur_frame_idx] = (nt > 0).long()
        return {"render": vb.color, "depth": vb.depth, "opacity": vb.opacity, "radii": vb.radii, "n_touched": nt}

    def map(self, current_window, prune=False, iters=1):
        if len(current_window) == 0:
            return
        if self.autograd_fallback:
            return MappingLoop.map(self, current_window, prune=prune, iters=iters)
        viewpoint_stack = [self.viewpoints[kf_idx] for kf_idx in current_window]
        frames_to_optimize = self.config["mapping"]["Training"]["pose_window"]
        cw = set(current_window)
        random_viewpoint_stack = [v for idx, v in self.viewpoints.items() if idx not in cw]
        pose_opt = self.keyframe_optimizers is not None
        gaussian_split = False
        self._span_cache = None      # the span arrays are shared by the spans of ONE map() call (between calls renders of other frames may
                                     # have re-assigned workspaces; re-making them once per call is 0.14 ms)
        it = -1
        while it + 1 < iters:
            it += 1
            # a span of regular iterations (no densification / opacity reset, no pose optimiser) is ONE host call
            n = 0
            if not prune and not pose_opt and self.span_calls and not self._has_stale():
                while it + n < iters and not self._is_special(self.iteration_count + n + 1):
                    n += 1
            if n > 0:
                self._ensure_state()
                c0 = self.iteration_count
                per = min(2, len(random_viewpoint_stack))
                run = self._run_span_ranks if self._parallel() else self._run_span
                # The draws and learning rates of an iteration cost the host ~15 us (torch.randperm, the lr schedule): for a
                # keyframe's 60 iterations that is ~1 ms before anything is enqueued -- with the GPU idle, because a span starts
                # behind a read-back.  The first 2 iterations therefore go out as a span of their own, the next 6 are drawn while
                # those run, the rest while these run (several spans = one span, iteration by iteration: bitwise,
                # tests/test_gpu_fused.py).
                picks, k0 = [], 0
                for k1 in ((2, 8, n) if (n > 12 and not self._parallel()) else (n,)):
                    part = []
                    for _ in range(k0, k1):          # the reference's draws, in its order (mapper.py:470)
                        part += torch.randperm(len(random_viewpoint_stack))[:2].tolist()
                    lrs = [float(self._xyz_group()["lr"]) if k == 0 else self.gaussians.lr_at(c0 + k) for k in range(k0, k1)]
                    run(viewpoint_stack, random_viewpoint_stack, part, lrs, 10.0, "window")
                    picks, k0 = part, k1
                self.iteration_count = c0 + n
                self.gaussians.update_learning_rate(self.iteration_count)
                self.last_used = list(viewpoint_stack) + [random_viewpoint_stack[k] for k in picks[len(picks) - per:]]
                gaussian_split = False               # (regular iterations: mapper.py:491 resets the flag every iteration)
                it += n - 1
                self._since_check += n
                if it == iters - 1 or self._since_check >= self.check_every:
                    self._txn_commit()               # (before anybody looks at what these iterations left behind)
                if it == iters - 1:
                    self.occ_aware_visibility = self._window_visibility(current_window, viewpoint_stack)
                    self._sync_moments()
                continue
            self.iteration_count += 1
            gaussian_split = False
            self._ensure_state()
            used = list(viewpoint_stack)
            for cam_idx in torch.randperm(len(random_viewpoint_stack))[:2].tolist():
                used.append(random_viewpoint_stack[cam_idx])
            if prune:
                # mapper.py:490-520: the whole iteration up to loss.backward(), then `return False` before the statistics,
                # optimizer.step() and zero_grad(): the gradients stay where they are
                self._step(used, adam=False, stats=False)
                self._txn_commit()
                self._stale_iso += 10.0
                if self._exp is not None:
                    self._exp.keep_stale([r for r in (self._exp.row_of(c) for c in used) if r is not None])
                self.occ_aware_visibility = self._window_visibility(current_window, viewpoint_stack)
                self._count_observations(current_window)
                self.last_used = used
                return False
            update_gaussian = self.iteration_count % self.gaussian_update_every == self.gaussian_update_offset
            reset = (self.iteration_count % self.gaussian_reset) == 0 and (not update_gaussian)
            special = update_gaussian or reset
            # regular iteration: everything in one host call; densify / reset iterations split around the torch-side surgery
            exp_stale = self._exp_stale()
            self._step(used, iso_weight=10.0, adam=not special,
                       exposure="none" if (special or pose_opt or exp_stale) else "window")
            if exp_stale and not (special or pose_opt):
                self._exposure_step(used)
            if special or pose_opt or it == iters - 1:
                self._txn_commit()                   # map surgery / pose steps / the caller consume this iteration's results
                with torch.no_grad():
                    if it == iters - 1 or update_gaussian:
                        # the reference rebuilds this dict every iteration (mapper.py:494-498); only the value that
                        # survives the call (or a change of N) is observable
                        self.occ_aware_visibility = self._window_visibility(current_window, viewpoint_stack)
                    if special or it == iters - 1:
                        self._sync_moments()
                    if update_gaussian:
                        if self._parallel():   # densification statistics are per rank: combine, then every rank takes
                            import torch.distributed as dist        # the identical (same-seed) decision
                            gm = self.gaussians
                            self.comm.all_reduce(gm.xyz_gradient_accum)
                            self.comm.all_reduce(gm.denom)
                            self.comm.all_reduce(gm.max_radii2D, op=dist.ReduceOp.MAX)
                        self.gaussians.densify_and_prune(self.opt_params.densify_grad_threshold, self.gaussian_th,
                                                         self.gaussian_extent, self.size_threshold)
                        gaussian_split = True
                        self._acc_key = None
                    if reset:
                        self.gaussians.reset_opacity_nonvisible(self._seen_by_any(used))
                        gaussian_split = True
                        self._step([], iso_weight=10.0, adam=True, skip=("opacity",), activate=False)
                    if special or pose_opt:
                        self._exposure_step(used)
                    if pose_opt:
                        for cam_idx in range(min(frames_to_optimize, len(current_window))):
                            if viewpoint_stack[cam_idx].uid == 0:
                                continue
                            update_pose(viewpoint_stack[cam_idx])
            self.gaussians.update_learning_rate(self.iteration_count)
            self.last_used = used
            self._tick()
        return gaussian_split

    def _window_visibility(self, current_window, viewpoint_stack):
        """occ_aware_visibility of mapper.py:494-498 from the last render of every window keyframe; with split views each
        keyframe was rendered by one rank: the masks are combined (max) so that every rank sees all of them."""
        if not (self._parallel() and self.split_views):
            # (one stack + one compare + one cast for the whole window instead of two launches per keyframe: the dict is rebuilt at
            # the end of every map() call; its values are rows of one [window, N] tensor)
            if not viewpoint_stack:
                return {}
            m = (torch.stack([self._views[c.uid].n_touched for c in viewpoint_stack]) > 0).long()
            return {kf: m[i] for i, kf in enumerate(current_window)}
        import torch.distributed as dist
        n = self.gaussians._xyz.shape[0]
        m = torch.zeros((len(viewpoint_stack), n), dtype=torch.int32, device=self.device)
        for i, c in enumerate(viewpoint_stack):
            if i % self.world == self.rank:
                m[i] = (self._views[c.uid].n_touched > 0).int()
        self.comm.all_reduce(m, op=dist.ReduceOp.MAX)
        return {kf: m[i].long() for i, kf in enumerate(current_window)}

    def _seen_by_any(self, used):
        """[radii > 0 of every view of the iteration] for reset_opacity_nonvisible (mapper.py:550-555); with split views the
        union over ALL ranks' views as one mask (the reference only uses the union)."""
        if not (self._parallel() and self.split_views):
            return [self._views[c.uid].radii > 0 for c in used]
        import torch.distributed as dist
        seen = torch.zeros(self.gaussians._xyz.shape[0], dtype=torch.int32, device=self.device)
        for c in self._local(used):
            seen |= (self._views[c.uid].radii > 0).int()
        self.comm.all_reduce(seen, op=dist.ReduceOp.MAX)
        return [seen.bool()]

    def _has_stale(self):
        """A prune pass left gradients behind that the next optimiser step must include (see the module docstring)."""
        return (not self._acc_clean) or self._stale_iso != 0.0 or self._exp_stale()

    def _exp_stale(self):
        """Rows of the CURRENT keyframe optimiser that carry a prune pass's exposure gradient (cameras outside it keep theirs,
        like the reference's never-zeroed .grad, but no optimiser ever looks at them again)."""
        return self._exp is not None and any(r in self._exp.stale_rows for r in self._exp_rows)

    def _is_special(self, count):
        update_gaussian = count % self.gaussian_update_every == self.gaussian_update_offset
        return update_gaussian or (count % self.gaussian_reset) == 0

    def _xyz_group(self):
        for g in self.gaussians.optimizer.param_groups:
            if g["name"] == "xyz":
                return g

    def final_refine(self, iters=26000, views_per_step=None):
        """mapper.py:656-708.  views_per_step = 1 (default; config mapping.final_refine_views_per_step): the reference's
        step -- ONE random view, Adam on all N -- which several ranks can only run replicated (deterministic kernels keep
        the replicas bit-identical; nothing to exchange, nothing gained).
        views_per_step = G > 1 ("world": one per rank): G DISTINCT random views per optimiser step and iters / G steps --
        the same number of renders, a G times larger batch per Adam step (SURVEY.md 8e "semantics caveat"; configs[4]).
        The G picks of a step come from ONE numpy stream that every rank draws identically; rank r renders picks r, r + world,
        ...; gradients meet in the ZeRO-1 exchange exactly like the views of a map() iteration.  A single process runs all
        G views of a step as one batched launch: the same optimisation up to fp32 summation order, which is how the PSNR of
        1 x iters against G x (iters / G) is measured on one GPU (scripts/refine_batch_psnr.py)."""
        if self.autograd_fallback:
            return MappingLoop.final_refine(self, iters)
        if views_per_step is None:
            views_per_step = self.config["mapping"].get("final_refine_views_per_step", 1)
        if views_per_step == "world":
            views_per_step = self.world
        views_per_step = max(1, min(int(views_per_step), len(self.viewpoints)))
        self._sync_moments()
        if views_per_step == 1:
            self._replicated += 1
            try:
                self._final_refine(iters)
                self._txn_commit()
                return
            finally:
                self._replicated -= 1
        split, self.split_views = self.split_views, True       # (a rank renders ITS picks of the step whatever map() does)
        try:
            self._final_refine_views(iters, views_per_step)
        finally:
            self.split_views = split

    def _final_refine_views(self, iters, views_per_step):
        # what a prune pass left behind is consumed by the reference's kind of step first (replicated: the stale sums are
        # identical on every rank and must enter the optimiser once)
        self._replicated += 1
        try:
            done = self._final_refine_stale(iters)
            self._txn_commit()                  # (a transaction never spans a change of the execution mode: its launches are re-issued as recorded)
        finally:
            self._replicated -= 1
        self._final_refine_multi(iters - done, views_per_step)
        self._txn_commit()
        self._sync_moments()

    def _final_refine_stale(self, iters):
        stack = list(self.viewpoints.values())
        done = 0
        while done < iters and self._has_stale():
            self.iteration_count += 1
            self._ensure_state()
            cam = stack[np.random.randint(0, len(stack))]
            self._step([cam], adam=True, stats=False, exposure="none")
            self._exposure_step([cam], only_rendered=True)
            self.gaussians.update_learning_rate(self.iteration_count)
            self.last_used = [cam]
            self._tick()
            done += 1
        return done

    def _final_refine_multi(self, iters, G):
        """ceil(iters / G) optimiser steps of G distinct random views each; `iteration_count` (the lr schedule's clock,
        mapper.py:705) advances by the number of renders."""
        stack = list(self.viewpoints.values())
        done = 0
        while done < iters:
            g = min(G, iters - done)
            self.iteration_count += g
            self._ensure_state()
            picks = []
            while len(picks) < g:                      # the reference's draw (mapper.py:668), repeated until distinct
                p = int(np.random.randint(0, len(stack)))
                if p not in picks:
                    picks.append(p)
            cams = [stack[p] for p in picks]
            self._step(cams, adam=True, stats=False, exposure="none")
            self._exposure_step(cams, only_rendered=True)
            self.gaussians.update_learning_rate(self.iteration_count)
            self.last_used = self._local(cams) or cams[:1]
            self._tick()
            done += g

    def _final_refine(self, iters=26000):
        stack = list(self.viewpoints.values())
        done = self._final_refine_stale(iters)           # what the last prune pass left behind goes into these steps
        while self.span_calls and done < iters:
            n = min(iters - done, 512)           # (the overflow check runs between chunks)
            self._ensure_state()
            c0 = self.iteration_count
            picks = [int(np.random.randint(0, len(stack))) for _ in range(n)]
            lrs = [float(self._xyz_group()["lr"])] + [self.gaussians.lr_at(c0 + k) for k in range(1, n)]
            self._run_span([], stack, picks, lrs, 0.0, "per_pick", stats=False)
            self.iteration_count = c0 + n
            self.gaussians.update_learning_rate(self.iteration_count)
            self.last_used = [stack[picks[-1]]]
            self._since_check += n
            if self._since_check >= self.check_every:
                self._txn_commit()
            done += n
        for _ in range(iters - done):
            self.iteration_count += 1
            self._ensure_state()
            cam = stack[np.random.randint(0, len(stack))]
            row = self._exp.row_of(cam) if self._exp is not None else None
            self._step([cam], adam=True, stats=False, exposure=[row] if row is not None else "none")
            self.gaussians.update_learning_rate(self.iteration_count)
            self.last_used = [cam]
            self._tick()

    def workspace_of(self, cam):
        """The object (per-camera buffers or a shared slot) whose saved / scratch blocks hold the LAST forward of `cam` --
        counters and histograms of bench.py / the session script read them."""
        j = getattr(self, "_last_pick_slots", {}).get(cam.uid)
        vb = self._views.get(cam.uid)
        if j is not None and (vb is None or vb.saved is None or not vb.ran):
            return self._slots[j]
        return vb

    # convenience for evaluation / tests
    @property
    def last_losses(self):
        return [self._views[c.uid].loss for c in getattr(self, "last_used", [])]

    def total_loss(self):
        return torch.stack([l[0] for l in self.last_losses]).sum()
