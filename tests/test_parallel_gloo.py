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
