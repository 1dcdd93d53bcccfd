"""ONE rank's share of a W-rank strong-scaling iteration, timed on one GPU: the loop is configured as rank 0 of W (views
dealt round-robin, ZeRO-1 rows of rank 0) with a stand-in Comm whose collectives move no data between devices
(reduce-scatter = take my slice, all-gather = leave the other slices as they are, all-reduce = nothing).  What it
gives is the per-iteration GPU + host time of a rank WITHOUT the xGMI time: an upper bound of the strong-scaling speed-up
of the reference's 12-view iteration.  (Parameters of the other ranks' rows never move here: timing only.)

    python scripts/multi_path_time.py [--kernels] [--gaussians N] [--scale-add 1.6] [--mode map|refine] [--out x.json]  ->  one JSON line per W

--mode refine: final_refine(views_per_step="world") (configs[4]: one random view per rank and optimiser step, mapper.py:617-710):
rank 0's share of a step = ONE view + its 1/W of the Adam pass."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from splat_slam_amd import synthetic as syn          # noqa: E402
from splat_slam_amd.fused import FusedMappingLoop     # noqa: E402


class NullComm:
    staged = False

    def __init__(self, world):
        self.world = world

    def all_reduce(self, t, op=None):
        pass

    def reduce_scatter(self, out, inp):
        out.copy_(inp[: out.numel()])                # rank 0's slice

    def all_gather(self, full, shard):
        pass


import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--kernels", action="store_true")
ap.add_argument("--gaussians", type=int, default=300000)
ap.add_argument("--scale-add", type=float, default=0.0)
ap.add_argument("--mode", default="map", choices=["map", "refine"])
ap.add_argument("--out", default=None)
A = ap.parse_args()
dev = torch.device("cuda:0")
intr = syn.INTRINSICS["metric"]
torch.manual_seed(43)
np.random.seed(43)
N = A.gaussians
params = syn.room_parameters(N, seed=43, device=dev)
if A.scale_add:
    params["scaling"] = params["scaling"] + A.scale_add
cams = syn.make_views(params, 16, intr, dev, seed=43)
rows = []
for world in (1, 2, 4, 8):
    loop = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev)
    loop.gaussians = syn.model_from_parameters(params, device=dev)
    loop.viewpoints = {c.uid: c for c in cams}
    loop.current_window = list(range(10))
    loop.build_keyframe_optimizers()
    loop.iteration_count = 50
    if world > 1:
        loop.set_parallel(world, 0, split_views=True, sync="zero1", comm=NullComm(world))
    ITERS = 80

    def run(iters):
        loop.iteration_count = 50
        if A.mode == "map":
            loop.map(loop.current_window, iters=iters)
        else:                                        # `iters` optimiser steps of `world` views each (rank 0 renders one of them)
            np.random.seed(7)
            loop.final_refine(iters=iters * world, views_per_step=world if world > 1 else 1)

    run(20)
    loop.check_overflow()
    loop.check_every = 1 << 30                       # (no capacity poll inside the timed window)
    torch.cuda.synchronize()
    tot = host = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        run(ITERS)
        h = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        if t < tot:
            tot, host = t, h
    kern = None
    if A.kernels:                      # HIP-event time per kernel kind (serialises the kernels: shares, not a sum)
        import ctypes as C
        import bench
        from splat_slam_amd import _native as nat
        lib = nat.lib()
        lib.sgr_profile_enable((1 << len(bench.KINDS)) - 1)
        run(40)
        torch.cuda.synchronize()
        ms, cnt = (C.c_float * len(bench.KINDS))(), (C.c_int64 * len(bench.KINDS))()
        lib.sgr_profile_read(ms, cnt)
        lib.sgr_profile_enable(0)
        kern = {bench.KINDS[i]: [round(float(ms[i]) / 40, 5), int(cnt[i]) // 40] for i in range(len(bench.KINDS)) if cnt[i]}
    row = {"world": world, "views_of_rank0_per_iteration": len(range(0, 12, world)) if A.mode == "map" else 1,
           "ms_per_iteration_rank0_no_collective_time": round(1e3 * tot / ITERS, 4), "host_enqueue_ms_per_iteration": round(1e3 * host / ITERS, 4),
           "bytes_reduce_scatter_plus_all_gather": 2 * 56 * N if world > 1 else 0,
           **({"kernel_ms_per_iteration_and_launches": kern} if kern else {})}
    rows.append(row)
    print(json.dumps(row), flush=True)
    del loop
    torch.cuda.empty_cache()
if A.out:
    os.makedirs(os.path.dirname(os.path.abspath(A.out)), exist_ok=True)
    base = rows[0]["ms_per_iteration_rank0_no_collective_time"]
    json.dump({"command": "python scripts/multi_path_time.py " + " ".join(sys.argv[1:]),
               "what": "rank 0's share of a W-rank strong-scaling ZeRO-1 %s on ONE MI355X, collectives replaced by a stand-in that moves no data "
                       "between devices: GPU + host time of a rank WITHOUT the xGMI time -- the expected 1 -> 8 curve the SCALE record "
                       "can be compared against (add 2 x 56 N / 8 bytes / 153 GB/s per exchange for all-pairs RCCL)" %
                       ("map() iteration (12 views dealt round-robin)" if A.mode == "map" else "final_refine step (one view per rank)"),
               "gaussians": N, "scale_add": A.scale_add, "mode": A.mode,
               "speedup_upper_bound": {str(r["world"]): round(base / r["ms_per_iteration_rank0_no_collective_time"] * (r["world"] if A.mode == "refine" else 1), 2)
                                       for r in rows},
               "rows": rows}, open(A.out, "w"), indent=1)
