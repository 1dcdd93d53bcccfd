"""-m gpu, needs >= 2 GPUs (skips cleanly on a 1-GPU box): the multi-GPU path on its REAL transport.

Every other multi-rank test of this repo moves its messages through host-staged gloo (two ranks sharing one GPU).  Here each
rank owns one GPU and the collectives are torch.distributed's "nccl" backend = RCCL over xGMI, exactly what
`bench.py --gpus N` runs under the driver: so that RCCL has executed this code in the test job before the scaling bench does.

  * the strong-scaling ZeRO-1 map() iteration (views dealt round-robin, ONE reduce-scatter -> sgr_gaussian_adam_shard on the
    rank's rows -> ONE in-place all-gather of the parameter buffer), with 2 ranks and with every GPU of the node:
    replicas bitwise, ZeRO-1 and all-reduce + replicated Adam equal up to RCCL's summation order, equal to the single-process loop
    up to summation order;
  * final_refine(views_per_step="world") (configs[4]'s loop, /root/reference/src/mapper.py:617-710, one random view per rank);
  * the in-place `all_gather_into_tensor` (a rank's shard IS its slot of the gathered buffer) and `reduce_scatter_tensor`
    of the flat 14-floats-per-Gaussian buffer against a host computation.

The reference has no distributed code: the contract tested here is this build's own.
"""
import os

import numpy as np
import pytest
import torch

NGPU = torch.cuda.device_count() if torch.cuda.is_available() else 0
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(NGPU < 2, reason="needs >= 2 GPUs: RCCL between ranks that own one GPU each")]
PARAMS = ["_xyz", "_features_dc", "_opacity", "_scaling", "_rotation"]
WORLDS = sorted({2, NGPU}) if NGPU >= 2 else [2]


def _scene(dev, n=2000, views=8, seed=21):
    from splat_slam_amd import synthetic as syn
    intr = syn.INTRINSICS["tiny"]
    params = syn.room_parameters(n, seed=seed, device=dev)
    params["scaling"] = params["scaling"] + 1.2
    cams = syn.make_views(params, views, intr, dev, seed=seed)
    return syn, params, cams


def _loop(dev, syn, params, cams, window):
    from splat_slam_amd.fused import FusedMappingLoop
    f = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev)
    f.gaussians = syn.model_from_parameters(params, device=dev)
    f.viewpoints = {c.uid: c for c in cams}
    f.current_window = list(window)
    f.build_keyframe_optimizers()
    f.iteration_count = 50
    return f


def _state(f):
    gm = f.gaussians
    st = {g["name"]: gm.optimizer.state[g["params"][0]] for g in gm.optimizer.param_groups}
    out = {k: getattr(gm, k).detach().cpu().clone() for k in PARAMS}
    out.update({"m_" + k: st[k]["exp_avg"].detach().cpu().clone() for k in ["xyz", "f_dc", "opacity", "scaling", "rotation"]})
    out.update({"v_" + k: st[k]["exp_avg_sq"].detach().cpu().clone() for k in ["xyz", "scaling"]})
    out["exposure"] = f._exp.param[:8].detach().cpu().clone()
    out["steps"] = torch.tensor([float(st[k]["step"]) for k in ["xyz", "f_dc", "opacity", "scaling", "rotation"]])
    out["overflow"] = torch.tensor([float(f.overflow_events)])
    return out


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    return dist, torch.device("cuda", rank)


def _map_worker(rank, world, port, out, sync, iters, reset_at):
    dist, dev = _init(rank, world, port)
    from splat_slam_amd.parallel import Comm
    syn, params, cams = _scene(dev)
    f = _loop(dev, syn, params, cams, [0, 1, 2, 3, 4, 5])
    f.set_parallel(world, rank, split_views=True, sync=sync, comm=Comm(staged=False))      # tensors go to RCCL as they are
    if reset_at:
        f.gaussian_reset = reset_at
    torch.manual_seed(3)
    f.map(f.current_window, iters=iters)
    f.map(f.current_window, iters=2)
    torch.cuda.synchronize()
    s = _state(f)
    s["occ"] = torch.stack([v for _, v in sorted(f.occ_aware_visibility.items())]).cpu()
    out[rank] = s
    dist.barrier()
    dist.destroy_process_group()


def _refine_worker(rank, world, port, out, views_per_step, iters):
    dist, dev = _init(rank, world, port)
    from splat_slam_amd.parallel import Comm
    syn, params, cams = _scene(dev)
    f = _loop(dev, syn, params, cams, [0, 1, 2, 3])
    f.set_parallel(world, rank, split_views=True, sync="zero1", comm=Comm(staged=False))
    np.random.seed(7)
    f.final_refine(iters=iters, views_per_step=views_per_step)
    torch.cuda.synchronize()
    out[rank] = _state(f)
    dist.barrier()
    dist.destroy_process_group()


def _collective_worker(rank, world, port, out, n):
    """The two collectives of an iteration on the flat buffer of parallel.Zero1Plan, against a host computation."""
    dist, dev = _init(rank, world, port)
    from splat_slam_amd.parallel import Comm, Zero1Plan
    plan = Zero1Plan(n, world, rank)
    comm = Comm(staged=False)
    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.randn(plan.total, generator=g)
    flat = mine.to(dev)
    shard = torch.empty(plan.shard, device=dev)
    comm.reduce_scatter(shard, flat)
    # every rank's contribution, regenerated on the host: the reduced shard must be their sum (RCCL's summation order across
    # ranks is not ours: compare with a tolerance of a few ulps of the partial sums)
    ref = torch.zeros(plan.total, dtype=torch.float64)
    for r in range(world):
        ref += torch.randn(plan.total, generator=torch.Generator().manual_seed(100 + r)).double()
    err = (shard.cpu().double() - ref[plan.lo:plan.hi]).abs().max().item()
    # in-place all-gather: this rank's shard is its own slot of the gathered buffer
    full = torch.full((plan.total,), -1.0, device=dev)
    full[plan.lo:plan.hi] = float(rank) + torch.arange(plan.shard, device=dev, dtype=torch.float32) / plan.shard
    comm.all_gather(full, full[plan.lo:plan.hi])
    exp = torch.cat([float(r) + torch.arange(plan.shard, dtype=torch.float32) / plan.shard for r in range(world)])
    out[rank] = {"rs_err": err, "ag_equal": bool(torch.equal(full.cpu(), exp))}
    dist.barrier()
    dist.destroy_process_group()


def _spawn(fn, world, *args):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = mp.Manager().dict()
    mp.spawn(fn, args=(world, port, out) + args, nprocs=world, join=True)
    return [out[r] for r in range(world)]


def _single(fn):
    dev = torch.device("cuda", 0)
    syn, params, cams = _scene(dev)
    return fn(dev, syn, params, cams)


def _close(multi, single, flips=0.01):
    lr = {"_xyz": 9.6e-4, "_features_dc": 2.5e-3, "_opacity": 0.05, "_scaling": 6e-3, "_rotation": 1e-3}
    for k, step in lr.items():        # same gradient sums up to fp32 summation order: a bounded few rounding-noise flips of Adam's sign
        d = (multi[k] - single[k]).abs()
        assert (d > 0.02 * step).float().mean().item() < flips, k
    assert (multi["exposure"] - single["exposure"]).abs().max().item() < 2e-3


@pytest.mark.parametrize("world", WORLDS)
def test_rccl_collectives_of_the_flat_buffer(world):
    res = _spawn(_collective_worker, world, 30011)
    for r in res:
        assert r["ag_equal"], "in-place all_gather_into_tensor"
        assert r["rs_err"] < 1e-5, r["rs_err"]


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("reset_at", [0, 53])
def test_rccl_strong_scaling_zero1_iteration(world, reset_at):
    z = _spawn(_map_worker, world, "zero1", 4, reset_at)
    for r in range(1, world):
        for k in z[0]:
            assert torch.equal(z[0][k], z[r][k]), ("ranks differ", r, k)
    assert z[0]["overflow"].item() == 0
    if not reset_at:
        assert z[0]["steps"].tolist() == [6.0] * 5
    a = _spawn(_map_worker, world, "allreduce", 4, reset_at)
    for k in z[0]:
        assert torch.equal(a[0][k], a[world - 1][k]), ("ranks differ (allreduce)", k)
    # RCCL's reduce-scatter and all-reduce may add the ranks' contributions in different orders: parameters agree up to that
    _close(z[0], a[0], flips=0.01)

    def single(dev, syn, params, cams):
        f = _loop(dev, syn, params, cams, [0, 1, 2, 3, 4, 5])
        if reset_at:
            f.gaussian_reset = reset_at
        torch.manual_seed(3)
        f.map(f.current_window, iters=4)
        f.map(f.current_window, iters=2)
        torch.cuda.synchronize()
        s = _state(f)
        s["occ"] = torch.stack([v for _, v in sorted(f.occ_aware_visibility.items())]).cpu()
        return s
    s = _single(single)
    _close(z[0], s)
    assert (z[0]["occ"] != s["occ"]).float().mean().item() < 0.01


@pytest.mark.parametrize("world", WORLDS)
def test_rccl_final_refine_one_view_per_rank(world):
    iters = 4 * world
    z = _spawn(_refine_worker, world, "world", iters)
    for r in range(1, world):
        for k in z[0]:
            assert torch.equal(z[0][k], z[r][k]), ("ranks differ", r, k)
    assert z[0]["steps"].tolist() == [4.0] * 5

    def single(dev, syn, params, cams):
        f = _loop(dev, syn, params, cams, [0, 1, 2, 3])
        np.random.seed(7)
        f.final_refine(iters=iters, views_per_step=world)
        torch.cuda.synchronize()
        return _state(f)
    _close(z[0], _single(single))
    # the reference's step (one view, replicated on every rank, nothing exchanged) keeps the replicas identical
    a = _spawn(_refine_worker, world, 1, 3)
    for k in a[0]:
        assert torch.equal(a[0][k], a[world - 1][k]), ("replicated refine: ranks differ", k)
