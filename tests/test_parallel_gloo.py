"""world_size-2 gloo tests of the multi-GPU exchange (splat_slam_amd/parallel.py) on CPU tensors."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeModel:
    def __init__(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        mk = lambda *s: torch.nn.Parameter(torch.randn(*s, generator=g))
        self._xyz, self._features_dc, self._features_rest = mk(n, 3), mk(n, 1, 3), torch.nn.Parameter(torch.zeros(n, 0, 3))
        self._opacity, self._scaling, self._rotation = mk(n, 1), mk(n, 3), mk(n, 4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from splat_slam_amd.parallel import GROUPS, GradientSync, ShardedAdamSync
    lrs = {"_xyz": 1e-3, "_features_dc": 2.5e-3, "_opacity": 0.05, "_scaling": 6e-3, "_rotation": 1e-3}
    # --- all-reduce path: every rank ends with the summed gradient
    m = _FakeModel(n, seed=1)                         # identical replicas
    g = torch.Generator().manual_seed(100 + rank)     # different views -> different gradients
    for name in GROUPS:
        p = getattr(m, name)
        if p.numel():
            p.grad = torch.randn(p.shape, generator=g)
    local = {name: getattr(m, name).grad.clone() for name in GROUPS if getattr(m, name).numel()}
    GradientSync(m, world).reduce()
    summed = {name: getattr(m, name).grad.clone() for name in local}
    # --- ZeRO-1 path: reduce-scatter + sharded Adam + all-gather == replicated Adam on the summed gradient
    m2 = _FakeModel(n, seed=1)
    for name, gr in local.items():
        getattr(m2, name).grad = gr.clone()
    sh = ShardedAdamSync(m2, world, rank, lrs)
    for _ in range(3):
        sh.step()
        for name, gr in local.items():
            getattr(m2, name).grad = gr.clone()
    out[rank] = (local, summed, {name: getattr(m2, name).detach().clone() for name in local})
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [5, 37])
def test_gradient_exchange_world_size_2(n):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    (l0, s0, p0), (l1, s1, p1) = out[0], out[1]
    lrs = {"_xyz": 1e-3, "_features_dc": 2.5e-3, "_opacity": 0.05, "_scaling": 6e-3, "_rotation": 1e-3}
    ref = _FakeModel(n, seed=1)
    opt = torch.optim.Adam([{"params": [getattr(ref, k)], "lr": lrs[k]} for k in l0], lr=0.0, eps=1e-15)
    for name in l0:
        assert torch.allclose(s0[name], l0[name] + l1[name], atol=1e-6)
        assert torch.equal(s0[name], s1[name])
    for _ in range(3):
        for name in l0:
            getattr(ref, name).grad = (l0[name] + l1[name]).clone()
        opt.step()
    for name in l0:
        assert torch.equal(p0[name], p1[name]), name                       # ranks agree bit-for-bit after all-gather
        assert torch.allclose(p0[name], getattr(ref, name).detach(), atol=1e-6, rtol=1e-5), name


# ---- the fused loop's ZeRO-1 exchange (parallel.Zero1Plan + Comm) with a torch stand-in for sgr_gaussian_adam_shard
def _torch_group_step(name, p, g_act, m, v, lr, step, iso_coef, b1=0.9, b2=0.999, eps=1e-15):
    """Rows of ONE group: chain rule from the gradient wrt the ACTIVATED input to the raw parameter, then Adam (what
    gaussian_adam_one does per Gaussian, csrc/sgr_aux.hip)."""
    if name == "opacity":
        sg = torch.sigmoid(p)
        g = g_act * sg * (1 - sg)
    elif name == "scaling":
        s = torch.exp(p)
        d = s - s.mean(dim=1, keepdim=True)
        sgn = torch.sign(d)
        g = (g_act + iso_coef * (sgn - sgn.sum(dim=1, keepdim=True) / 3.0)) * s
    elif name == "rotation":
        nrm = p.norm(dim=1, keepdim=True)
        y = p / nrm
        g = (g_act - y * (y * g_act).sum(dim=1, keepdim=True)) / nrm
    else:
        g = g_act
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.sub_((lr / (1 - b1 ** step)) * m / (v.sqrt() / (1 - b2 ** step) ** 0.5 + eps))


def _zero1_worker(rank, world, port, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from splat_slam_amd.parallel import WIDTHS, Comm, Zero1Plan
    plan, comm = Zero1Plan(n, world, rank), Comm()
    shapes = {"xyz": (n, 3), "f_dc": (n, 3), "opacity": (n, 1), "scaling": (n, 3), "rotation": (n, 4)}
    lrs = {"xyz": 1e-3, "f_dc": 2.5e-3, "opacity": 0.05, "scaling": 6e-3, "rotation": 1e-3}
    g0 = torch.Generator().manual_seed(1)
    flat_p, flat_m, flat_v = torch.zeros(plan.total), torch.zeros(plan.total), torch.zeros(plan.total)
    for name, _ in WIDTHS:
        plan.view(flat_p, name, shapes[name]).copy_(torch.randn(shapes[name], generator=g0))       # identical replicas
    start = flat_p.clone()
    shard = torch.zeros(plan.shard)
    locals_ = []
    for step in (1, 2, 3):
        gl = torch.Generator().manual_seed(100 * step + rank)          # this rank's views -> its own gradient sums
        flat_g = torch.zeros(plan.total)
        for name, _ in WIDTHS:
            plan.view(flat_g, name, shapes[name]).copy_(torch.randn(shapes[name], generator=gl))
        locals_.append(flat_g.clone())
        comm.reduce_scatter(shard, flat_g)
        for name, w in WIDTHS:
            r0, r1 = plan.rows[name]
            if r1 <= r0:
                continue
            base = plan.grad_base_offset(name)
            g_rows = shard[base + w * r0: base + w * r1].view(r1 - r0, w)      # the pointer arithmetic the HIP call is handed
            sl = lambda buf: plan.view(buf, name, shapes[name])[r0:r1]
            _torch_group_step(name, sl(flat_p), g_rows, sl(flat_m), sl(flat_v), lrs[name], step, 10.0 / (3.0 * n))
        comm.all_gather(flat_p, flat_p[plan.lo:plan.hi].clone())
    comm.all_gather(flat_m, flat_m[plan.lo:plan.hi].clone())              # _sync_moments
    out[rank] = (start, locals_, flat_p.clone(), flat_m.clone(), {k: v for k, v in plan.rows.items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 50, 1001])
def test_zero1_plan_exchange_world_size_2_equals_replicated_adam(n):
    from splat_slam_amd.parallel import WIDTHS, Zero1Plan
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_zero1_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    (start, l0, p0, m0, rows0), (_, l1, p1, m1, rows1) = out[0], out[1]
    assert torch.equal(p0, p1) and torch.equal(m0, m1)                     # bitwise-equal replicas after the all-gathers
    for name, _ in WIDTHS:                                                 # the two ranks' rows tile [0, n) exactly
        (a0, b0), (a1, b1) = rows0[name], rows1[name]
        assert (b0 - a0) + (b1 - a1) == n and (a1 == b0 or b0 == a0 or b1 == a1)
    # single process: the summed gradient, the same per-group step on ALL rows
    plan = Zero1Plan(n, 1, 0)
    ref = Zero1Plan(n, world, 0)
    shapes = {"xyz": (n, 3), "f_dc": (n, 3), "opacity": (n, 1), "scaling": (n, 3), "rotation": (n, 4)}
    lrs = {"xyz": 1e-3, "f_dc": 2.5e-3, "opacity": 0.05, "scaling": 6e-3, "rotation": 1e-3}
    P = {name: ref.view(start.clone(), name, shapes[name]).clone() for name, _ in WIDTHS}
    M = {name: torch.zeros(shapes[name]) for name, _ in WIDTHS}
    V = {name: torch.zeros(shapes[name]) for name, _ in WIDTHS}
    for step in (1, 2, 3):
        for name, _ in WIDTHS:
            g = ref.view(l0[step - 1], name, shapes[name]) + ref.view(l1[step - 1], name, shapes[name])
            _torch_group_step(name, P[name], g, M[name], V[name], lrs[name], step, 10.0 / (3.0 * n))
    for name, _ in WIDTHS:
        assert torch.equal(ref.view(p0, name, shapes[name]), P[name]), name
        assert torch.equal(ref.view(m0, name, shapes[name]), M[name]), name
    assert plan.rows["xyz"] == (0, n)


@pytest.mark.parametrize("world,n", [(4, 7), (4, 1001), (8, 5), (8, 1001)])
def test_zero1_plan_exchange_world_size_4_and_8(world, n):
    """the ZeRO-1 plan beyond two ranks, on map sizes that are no multiple of 12 x world (and smaller than the
    world: ranks that own no row of a group).  Replicas bitwise equal after the all-gathers, every group's rows tile [0, n) over the
    ranks in rank order, and the result equals the single-process step on the summed gradient up to the summation order of `world`
    fp32 terms (two ranks: bitwise, the test above)."""
    from splat_slam_amd.parallel import WIDTHS, Zero1Plan
    out = mp.Manager().dict()
    mp.spawn(_zero1_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    res = [out[r] for r in range(world)]
    start = res[0][0]
    for r in range(1, world):
        assert torch.equal(res[r][2], res[0][2]) and torch.equal(res[r][3], res[0][3]), r
    for name, _ in WIDTHS:
        edge = 0
        for r in range(world):
            a, b = res[r][4][name]
            assert b >= a
            if b > a:
                assert a == edge, (name, r, a, edge)
                edge = b
        assert edge == n, (name, edge)
    ref = Zero1Plan(n, world, 0)
    shapes = {"xyz": (n, 3), "f_dc": (n, 3), "opacity": (n, 1), "scaling": (n, 3), "rotation": (n, 4)}
    lrs = {"xyz": 1e-3, "f_dc": 2.5e-3, "opacity": 0.05, "scaling": 6e-3, "rotation": 1e-3}
    P = {name: ref.view(start.clone(), name, shapes[name]).clone().double() for name, _ in WIDTHS}
    M = {name: torch.zeros(shapes[name], dtype=torch.float64) for name, _ in WIDTHS}
    V = {name: torch.zeros(shapes[name], dtype=torch.float64) for name, _ in WIDTHS}
    for step in (1, 2, 3):
        for name, _ in WIDTHS:
            g = sum(ref.view(res[r][1][step - 1], name, shapes[name]).double() for r in range(world))
            _torch_group_step(name, P[name], g, M[name], V[name], lrs[name], step, 10.0 / (3.0 * n))
    for name, _ in WIDTHS:
        assert torch.allclose(ref.view(res[0][2], name, shapes[name]).double(), P[name], atol=2e-6, rtol=1e-5), name
        assert torch.allclose(ref.view(res[0][3], name, shapes[name]).double(), M[name], atol=2e-6, rtol=1e-5), name
