"""GPU box: where the HOST time of a mapping session goes (cProfile over keyframes of the converged phase of the
configs[1]-shaped session).   python scripts/profile_session.py [--warm 100] [--frames 30]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--warm", type=int, default=100)
ap.add_argument("--frames", type=int, default=30)
a = ap.parse_args()
from splat_slam_amd import synthetic as syn  # noqa: E402
from splat_slam_amd.fused import FusedMappingLoop  # noqa: E402
from splat_slam_amd.session import MappingSession  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(43)
np.random.seed(43)
intr = syn.INTRINSICS["metric"]
n = a.warm + a.frames
frames = syn.keyframe_stream(160, intr, dev, n_world=400000, seed=43, sweep_deg=360.0 * 159 / 160)[:n]
loop = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev)
sess = MappingSession(loop, intr)
for f in frames[: a.warm]:
    sess.process(*f)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for f in frames[a.warm:]:
    sess.process(*f)
pr.disable()
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("frames %d  N %d  host %.1f ms/frame  wall %.1f ms/frame" % (a.frames, loop.gaussians.get_xyz.shape[0], 1e3 * host / a.frames, 1e3 * wall / a.frames))
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(35)
    print(s.getvalue()[:9000])
