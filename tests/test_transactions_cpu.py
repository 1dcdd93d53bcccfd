"""CPU: the never-drop transaction protocol of FusedMappingLoop (splat_slam_amd/fused.py: _txn_begin / _txn_do / _txn_commit) without a
GPU -- the launches of a transaction are stood in for by closures that mutate the optimisation state in place, the header read-back by
a scripted answer.  What is checked is the PROTOCOL: snapshot, restore, replay exactly once per correction, Python-side counters, and
-- world_size 2 over gloo -- that both ranks replay when only one of them saw an overflow (the flag is all-reduced; a rank that went on
alone would leave its peer in the journal's collectives).  The numerics of a real replay are the -m gpu tests (tests/test_gpu_round5.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _loop(n=64, seed=3):
    from splat_slam_amd import synthetic as syn
    from splat_slam_amd.fused import FusedMappingLoop
    g = torch.Generator().manual_seed(seed)
    params = dict(xyz=torch.randn(n, 3, generator=g), f_dc=torch.randn(n, 1, 3, generator=g), opacity=torch.randn(n, 1, generator=g),
                  scaling=torch.randn(n, 3, generator=g), rotation=torch.randn(n, 4, generator=g))
    f = FusedMappingLoop(syn.DEFAULT_CONFIG, device="cpu", knn_fn=lambda p: torch.ones(p.shape[0]))
    f.gaussians = syn.model_from_parameters(params, device="cpu", knn_fn=lambda p: torch.ones(p.shape[0]))
    for grp in f.gaussians.optimizer.param_groups:        # Adam state as _ensure_state would create it
        p = grp["params"][0]
        f.gaussians.optimizer.state[p] = {"step": torch.tensor(0.0), "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
    return f


def _fake_iteration(f, log):
    """What a span does to the state, in place: parameters, moments, statistics, step counters."""
    gm = f.gaussians
    with torch.no_grad():
        for grp in gm.optimizer.param_groups:
            p = grp["params"][0]
            if p.numel() == 0:
                continue
            st = gm.optimizer.state[p]
            st["exp_avg"].mul_(0.9).add_(0.1)
            st["exp_avg_sq"].mul_(0.999).add_(0.001)
            p.data.sub_(0.01 * st["exp_avg"] / (st["exp_avg_sq"].sqrt() + 1e-15))
            st["step"] += 1
        gm.xyz_gradient_accum.add_(1.0)
        gm.denom.add_(1.0)
    log.append("run")


def _state(f):
    gm = f.gaussians
    out = {}
    for grp in gm.optimizer.param_groups:
        p = grp["params"][0]
        if p.numel():
            st = gm.optimizer.state[p]
            out[grp["name"]] = (p.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), float(st["step"]))
    out["accum"] = gm.xyz_gradient_accum.clone()
    return out


def _equal(a, b):
    for k in a:
        if isinstance(a[k], tuple):
            assert all(torch.equal(x, y) if torch.is_tensor(x) else x == y for x, y in zip(a[k], b[k])), k
        else:
            assert torch.equal(a[k], b[k]), k


def test_commit_without_overflow_runs_everything_once():
    f, log = _loop(), []
    f._read_overflows = lambda: []
    for _ in range(3):
        f._txn_do(lambda: _fake_iteration(f, log))
    assert f._txn is not None and len(f._txn.journal) == 3
    assert f._txn_commit() == [] and f._txn is None and log == ["run"] * 3 and f.replayed_transactions == 0


def test_overflow_restores_and_replays_the_journal_once():
    ref, rlog = _loop(), []
    ref._read_overflows = lambda: []
    for _ in range(3):
        ref._txn_do(lambda: _fake_iteration(ref, rlog))
    ref._txn_commit()
    f, log = _loop(), []
    answers = [["cam 7"], []]                      # the first look finds a truncated forward, the look after the replay is clean
    f._read_overflows = lambda: answers.pop(0)
    f._stale_iso = 10.0
    for _ in range(3):
        f._txn_do(lambda: _fake_iteration(f, log))
    f._stale_iso = 0.0                             # (consumed by a step of the transaction: Python-side state is part of the snapshot)
    assert f._txn_commit() == ["cam 7"]
    assert log == ["run"] * 6 and f.replayed_transactions == 1 and f._txn is None and not answers
    assert f._stale_iso == 10.0                    # restored (the replayed closures of a real loop consume it again)
    _equal(_state(f), _state(ref))                 # three iterations' worth of state, not six


def test_capacity_that_never_fits_raises_instead_of_looping():
    import pytest
    f, log = _loop(), []
    f._read_overflows = lambda: ["cam 1"]
    f._txn_do(lambda: _fake_iteration(f, log))
    with pytest.raises(RuntimeError, match="still overflows"):
        f._txn_commit()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from splat_slam_amd.parallel import Comm
    f, log = _loop(), []
    f.set_parallel(world, rank, split_views=True, sync="allreduce", comm=Comm())
    answers = [["cam 3"], []] if rank == world - 1 else [[], []]          # only the LAST rank sees a truncated forward

    def span():
        _fake_iteration(f, log)
        t = torch.ones(4)
        f.comm.all_reduce(t)                        # the journal's own collective: a rank replaying alone would hang here
        assert float(t[0]) == world
    f._read_overflows = lambda: answers.pop(0)
    f._txn_do(span)
    f._txn_do(span)
    first = f._txn_commit()
    out[rank] = (len(log), f.replayed_transactions, first, _state(f))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 4, 8])
def test_all_ranks_replay_together_over_gloo(world):
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        n, rep, first, st = out[r]
        assert n == 4 and rep == 1, (r, n, rep)                 # two launches, replayed once, on EVERY rank
        assert first == (["cam 3"] if r == world - 1 else [])
        _equal(st, out[0][3])
    s0 = out[0][3]
    ref, rlog = _loop(), []
    ref._read_overflows = lambda: []
    for _ in range(2):
        ref._txn_do(lambda: _fake_iteration(ref, rlog))
    ref._txn_commit()
    _equal(s0, _state(ref))


def test_a_truncated_forward_in_the_middle_of_a_span_is_still_seen_at_the_check():
    """SavedHeader.overflow is rewritten by every forward, a span puts ~52 forwards through one workspace between
    two checks and a slot renders a different camera each time.  K2 therefore also keeps a STICKY count of truncated forwards
    (header word 12) and the largest pair count demanded (word 13); _apply_headers reports a workspace whose count CHANGED since the
    previous check even though its last forward was fine, and sizes the replay by the worst forward."""
    import numpy as np
    from splat_slam_amd.fused import _Slot
    f = _loop()
    f._cap, f.capacity_floor, f.max_pairs = 1 << 16, 1 << 16, 1 << 30
    sl = _Slot()
    sl.pairs, sl.estimated = 100, False

    def header(last_R, last_ov, events, max_R, longest=7):
        w = np.zeros(16, dtype=np.uint32)
        w[0], w[1], w[10], w[12], w[13] = last_R, last_ov, longest, events, max_R
        return torch.from_numpy(w.view(np.uint8).copy())

    todo = [(("slot", 0), sl)]
    assert f._apply_headers(todo, header(100, 0, 0, 100)) == []                       # nothing happened
    got = f._apply_headers(todo, header(120, 0, 1, 90000))                            # a middle forward overflowed, the last one did not
    assert got == [("slot", 0)] and f.overflow_events == 1
    assert sl.pairs == 90000 and f._cap >= 2 * 90000                                  # the replay is sized by the worst forward
    assert f._apply_headers(todo, header(120, 0, 1, 90000)) == []                     # acknowledged: the same count is not news
    assert f._apply_headers(todo, header(95000, 1, 2, 95000)) == [("slot", 0)]        # the classic case: the last forward itself
