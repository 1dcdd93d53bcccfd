"""GPU box: where the initialisation keyframe's time goes (1050 single-view iterations at N <= 35 k took 392-432 ms,
several times their kernel time).  Runs the init keyframe of the bench session twice (the second is the measurement), with
  * synchronised wall time of every map-surgery call (densify_and_prune, reset_opacity) and of the seeding (add_next_kf),
  * synchronised wall time of every span / single step, their iteration counts,
  * HIP-event totals per kernel kind over the whole call (sgr_profile_enable: adds its own overhead, so in a THIRD run).
    python scripts/profile_init.py [--out x.json]"""
import argparse
import collections
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
a = ap.parse_args()
from splat_slam_amd import synthetic as syn  # noqa: E402
from splat_slam_amd.fused import FusedMappingLoop  # noqa: E402
from splat_slam_amd.session import MappingSession  # noqa: E402

dev = torch.device("cuda:0")
intr = syn.INTRINSICS["metric"]
torch.manual_seed(43)
np.random.seed(43)
frames = syn.keyframe_stream(2, intr, dev, n_world=400000, seed=43, sweep_deg=360.0 / 160)


def run(sync_parts, kernel_events):
    torch.manual_seed(43)
    np.random.seed(43)
    loop = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev)
    sess = MappingSession(loop, intr)
    parts = collections.defaultdict(lambda: [0.0, 0])

    def wrap(obj, name, label, count=None):
        fn = getattr(obj, name)

        def inner(*args, **kw):
            if sync_parts:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn(*args, **kw)
            if sync_parts:
                torch.cuda.synchronize()
            p = parts[label]
            p[0] += time.perf_counter() - t0
            p[1] += count(args, kw) if count else 1
            return r
        setattr(obj, name, inner)

    gm_cls_patches = []
    wrap(loop, "add_next_kf", "add_next_kf (seeding, knn)")
    wrap(loop, "_run_span", "_run_span (regular iterations, ONE host call)", count=lambda args, kw: len(args[3]))
    wrap(loop, "_step", "_step (single iterations around surgery)")
    wrap(loop, "_txn_commit", "_txn_commit (capacity check: one sync)")
    wrap(loop, "_ensure_state", "_ensure_state")
    if kernel_events:
        loop.lib.sgr_profile_enable(0x7f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # densify_and_prune / reset_opacity live on the model, which add_next_kf creates: patch the class for the duration of the call
    from splat_slam_amd import gaussian_model as gmod
    saved = {}
    for name in ("densify_and_prune", "reset_opacity"):
        orig = getattr(gmod.GaussianModel, name)
        saved[name] = orig

        def timed(self, *args, __orig=orig, __name=name, **kw):
            if sync_parts:
                torch.cuda.synchronize()
            t = time.perf_counter()
            r = __orig(self, *args, **kw)
            if sync_parts:
                torch.cuda.synchronize()
            parts[__name][0] += time.perf_counter() - t
            parts[__name][1] += 1
            return r
        setattr(gmod.GaussianModel, name, timed)
    try:
        st = sess.process(*frames[0])
    finally:
        for name, orig in saved.items():
            setattr(gmod.GaussianModel, name, orig)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    out = {"status": st, "wall_ms": round(1e3 * wall, 2), "gaussians_after_init": int(loop.gaussians.get_xyz.shape[0]),
           "parts_ms": {k: {"ms": round(1e3 * v[0], 2), "calls_or_iterations": v[1]} for k, v in parts.items()}}
    if kernel_events:
        ms, cnt = (C.c_float * 7)(), (C.c_int64 * 7)()
        loop.lib.sgr_profile_read(ms, cnt)
        loop.lib.sgr_profile_enable(0)
        names = ["preprocess_fwd", "tile_scan", "scatter", "blend_fused", "blend_fwd", "blend_bwd", "preprocess_bwd_incl_optimiser"]
        out["kernel_ms_total"] = {n: {"ms": round(float(ms[i]), 2), "launches": int(cnt[i])} for i, n in enumerate(names) if int(cnt[i])}
        out["kernel_ms_sum"] = round(sum(float(ms[i]) for i in range(7)), 2)
    return out


run(False, False)                        # warm-up: allocator, lazy code loading
res = {"plain": run(False, False), "synchronised_parts": run(True, False), "kernel_events": run(False, True)}
print(json.dumps(res, indent=1))
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
