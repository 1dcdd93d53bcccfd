"""Multi-GPU map refinement: view-parallel rendering with one gradient exchange per optimiser step.

The reference has no distributed code at all (grep torch.distributed|nccl -> nothing); this is new design for one
node of 8 MI355X over xGMI.  Alpha compositing is order dependent per pixel, so Gaussians cannot be sharded for
rendering; what shards is the VIEW set of one optimisation step (its losses are summed, src/mapper.py:450,480,490):
every rank renders its own views of a replicated map, then gradients are summed.

Exchange: the 14 floats per Gaussian (xyz3 f_dc3 opacity1 scale3 rot4) are packed into ONE flat fp32 bucket
(56 B x N: 16.8 MB at 300k, 84 MB at 1.5M) and reduced with a single RCCL call -- one large collective per step is the
right shape for xGMI's 7 point-to-point links (per-link bound), not many small per-tensor ones.
`GradientSync`   : all-reduce(sum) -> every rank runs the identical Adam step (replicated optimiser state).
`ShardedAdamSync`: reduce-scatter(sum) -> fused Adam on this rank's 1/G slice -> all-gather of the parameters
                   (ZeRO-1 layout: same bytes on the wire, optimiser state and update sharded G ways).
Both work on the gloo backend (CPU tensors) for the world_size-2 tests.
"""
import torch
import torch.distributed as dist

GROUPS = ["_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"]


def _params(gm):
    return [getattr(gm, name) for name in GROUPS if getattr(gm, name).numel() > 0]


class GradientSync:
    def __init__(self, gaussians, world_size, group=None):
        self.gm = gaussians
        self.world = world_size
        self.group = group
        self._bucket = None

    def reduce(self):
        """Sum the gradients of all ranks in place (call between backward() and optimizer.step())."""
        if self.world <= 1:
            return
        ps = _params(self.gm)
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in ps]
        total = sum(g.numel() for g in grads)
        if self._bucket is None or self._bucket.numel() != total or self._bucket.device != grads[0].device:
            self._bucket = torch.empty(total, dtype=torch.float32, device=grads[0].device)
        off = 0
        for g in grads:
            self._bucket[off:off + g.numel()].copy_(g.reshape(-1))
            off += g.numel()
        dist.all_reduce(self._bucket, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for p, g in zip(ps, grads):
            red = self._bucket[off:off + g.numel()].view_as(g)
            if p.grad is None:
                p.grad = red.clone()
            else:
                p.grad.copy_(red)
            off += g.numel()


class ShardedAdamSync:
    """reduce-scatter -> Adam on the local slice -> all-gather.  Owns flat (param, m, v) slabs; the six model tensors
    are refreshed from the gathered slab after every step.  lr differs per group, so the slice update is applied per
    group-segment intersection."""

    def __init__(self, gaussians, world_size, rank, lrs, betas=(0.9, 0.999), eps=1e-15, group=None, adam_fn=None):
        self.gm, self.world, self.rank, self.group = gaussians, world_size, rank, group
        self.lrs = dict(lrs)          # group attr name -> lr
        self.betas, self.eps = betas, eps
        self.step_count = 0
        self.adam_fn = adam_fn
        self._layout()

    def _layout(self):
        ps = [(n, getattr(self.gm, n)) for n in GROUPS if getattr(self.gm, n).numel() > 0]
        self.segments = []
        off = 0
        for n, p in ps:
            self.segments.append((n, off, off + p.numel()))
            off += p.numel()
        self.total = off
        self.padded = (off + self.world - 1) // self.world * self.world
        self.shard = self.padded // self.world
        dev = ps[0][1].device
        self.flat_grad = torch.zeros(self.padded, dtype=torch.float32, device=dev)
        self.flat_param = torch.zeros(self.padded, dtype=torch.float32, device=dev)
        for (n, a, b), (_, p) in zip(self.segments, ps):
            self.flat_param[a:b].copy_(p.detach().reshape(-1))
        self.my_grad = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.m = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.shard, dtype=torch.float32, device=dev)

    def set_lr(self, name, lr):
        self.lrs[name] = lr

    @torch.no_grad()
    def step(self):
        for n, a, b in self.segments:
            g = getattr(self.gm, n).grad
            if g is None:
                self.flat_grad[a:b].zero_()
            else:
                self.flat_grad[a:b].copy_(g.reshape(-1))
        if self.world > 1:
            dist.reduce_scatter_tensor(self.my_grad, self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self.my_grad.copy_(self.flat_grad[: self.shard])
        self.step_count += 1
        lo, hi = self.rank * self.shard, (self.rank + 1) * self.shard
        mine = self.flat_param[lo:hi]
        for n, a, b in self.segments:
            s, e = max(a, lo), min(b, hi)
            if s >= e:
                continue
            sl = slice(s - lo, e - lo)
            self._adam(mine[sl], self.my_grad[sl], self.m[sl], self.v[sl], self.lrs[n])
        if self.world > 1:
            dist.all_gather_into_tensor(self.flat_param, mine.clone(), group=self.group)
        for n, a, b in self.segments:
            p = getattr(self.gm, n)
            p.data.copy_(self.flat_param[a:b].view_as(p))
            p.grad = None

    def _adam(self, p, g, m, v, lr):
        if self.adam_fn is not None and p.is_cuda:
            self.adam_fn(p, g, m, v, lr, self.betas[0], self.betas[1], self.eps, self.step_count)
            return
        b1, b2 = self.betas
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** self.step_count
        bc2 = 1 - b2 ** self.step_count
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(self.eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


# ----------------------------------------------------------------------------------------------------------------------
# Layout + collectives of the fused loop's multi-GPU iteration (splat_slam_amd/fused.py).  Device agnostic: the gloo tests
# drive it with CPU tensors and a torch stand-in for the HIP Adam kernel.
# ----------------------------------------------------------------------------------------------------------------------
WIDTHS = (("xyz", 3), ("f_dc", 3), ("opacity", 1), ("scaling", 3), ("rotation", 4))     # floats per Gaussian, sink order


class Comm:
    """The three collectives of an iteration.  `staged=True` moves every message through host memory (gloo): that is how
    two ranks that share ONE GPU exchange data in the tests; on a real node the tensors go to RCCL as they are."""

    def __init__(self, group=None, staged=False):
        self.group, self.staged = group, staged

    def all_reduce(self, t, op=None):
        op = dist.ReduceOp.SUM if op is None else op
        if not self.staged:
            dist.all_reduce(t, op=op, group=self.group)
            return
        c = t.detach().cpu()
        dist.all_reduce(c, op=op, group=self.group)
        t.copy_(c.to(t.device))

    def reduce_scatter(self, out, inp):
        if not self.staged:
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group)
            return
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.reduce_scatter_tensor(o, inp.detach().cpu(), op=dist.ReduceOp.SUM, group=self.group)
        out.copy_(o.to(out.device))

    def all_gather(self, full, shard):
        """full[r*S:(r+1)*S] <- rank r's shard; `shard` may be this rank's own slot of `full` (in place on RCCL)."""
        if not self.staged:
            dist.all_gather_into_tensor(full, shard, group=self.group)
            return
        f = torch.empty(full.shape, dtype=full.dtype)
        dist.all_gather_into_tensor(f, shard.detach().cpu().clone(), group=self.group)
        full.copy_(f.to(full.device))


class Zero1Plan:
    """Which rows of which parameter group a rank owns when ONE flat buffer is reduce-scattered (ZeRO-1).

    The flat buffer is group-major: group k (width w_k floats per Gaussian) occupies [off_k, off_k + n_pad * w_k) with
    off_k = n_pad * (w_0 + ... + w_{k-1}).  n_pad = N rounded up to a multiple of 12 * world: then every boundary
    r * S of the equal shards S = 14 * n_pad / world falls on a row boundary of whatever group it lands in (12 = lcm of
    the widths; 14 * 12 = 168 is divisible by 3 and 4), so a shard is at most a few whole-row segments: ONE
    reduce-scatter and ONE all-gather per iteration, no packing pass."""

    def __init__(self, n, world, rank):
        self.n, self.world, self.rank = n, world, rank
        q = 12 * world
        self.n_pad = (max(n, 1) + q - 1) // q * q
        self.total = 14 * self.n_pad
        self.shard = self.total // world
        self.offsets, off = {}, 0
        for name, w in WIDTHS:
            self.offsets[name] = off
            off += self.n_pad * w
        self.lo, self.hi = rank * self.shard, (rank + 1) * self.shard
        self.rows = {}            # name -> (row0, row1) owned by this rank, clipped to the real N
        for name, w in WIDTHS:
            a, b = max(self.lo, self.offsets[name]), min(self.hi, self.offsets[name] + self.n_pad * w)
            if a >= b:
                self.rows[name] = (0, 0)
                continue
            assert (a - self.offsets[name]) % w == 0 and (b - self.offsets[name]) % w == 0
            r0, r1 = (a - self.offsets[name]) // w, (b - self.offsets[name]) // w
            self.rows[name] = (min(r0, n), min(r1, n))

    def view(self, flat, name, shape):
        """The [N, ...] tensor of group `name` inside a flat buffer of this layout."""
        w = dict(WIDTHS)[name]
        return flat[self.offsets[name]: self.offsets[name] + self.n * w].view(shape)

    def grad_base_offset(self, name):
        """Element offset (may be negative) from the start of the rank's gradient shard to the VIRTUAL row 0 of group `name`:
        shard[grad_base_offset + w * i] is the reduced gradient of Gaussian i for every i in rows[name]."""
        return self.offsets[name] - self.lo
