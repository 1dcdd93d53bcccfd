"""GPU box: cProfile of FusedMappingLoop._prepare_span over the second half of the bench session (after a 12-frame throw-away session,
like bench.py's session leg).   python scripts/micro/prepare_span_profile.py [frames=40]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from splat_slam_amd import synthetic as syn
from splat_slam_amd.fused import FusedMappingLoop
from splat_slam_amd.session import MappingSession
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0"); intr = syn.INTRINSICS["metric"]
def session(k):
    torch.manual_seed(43); np.random.seed(43)
    frames = syn.keyframe_stream(n, intr, dev, n_world=400000, seed=43, sweep_deg=360.0 * (n - 1) / 160)[:k]
    loop = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev); sess = MappingSession(loop, intr)
    return frames, loop, sess
frames, loop, sess = session(12)
for f in frames: sess.process(*f)
torch.cuda.synchronize(); del loop, sess
frames, loop, sess = session(n)
pr = cProfile.Profile()
orig = FusedMappingLoop._prepare_span
tot = [0.0, 0]
def wrapped(self, *a, **k):
    t = time.perf_counter(); pr.enable()
    try:
        return orig(self, *a, **k)
    finally:
        pr.disable(); tot[0] += time.perf_counter() - t; tot[1] += 1
for i, f in enumerate(frames):
    if i == n // 2: FusedMappingLoop._prepare_span = wrapped
    sess.process(*f)
print("calls", tot[1], "total ms", round(1e3 * tot[0], 2))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
