"""GPU box: what _txn_begin (the snapshot in front of a keyframe's first span) spends its host time on in a YOUNG session (the map
grows at every keyframe): python scripts/micro/txn_begin_parts.py [frames=40]"""
import collections, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from splat_slam_amd import synthetic as syn
from splat_slam_amd.fused import FusedMappingLoop
from splat_slam_amd.session import MappingSession
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0"); intr = syn.INTRINSICS["metric"]
torch.manual_seed(43); np.random.seed(43)
frames = syn.keyframe_stream(n, intr, dev, n_world=400000, seed=43, sweep_deg=360.0 * (n - 1) / 160)
loop = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev); sess = MappingSession(loop, intr)
T, C = collections.Counter(), collections.Counter()
inside = [False]
def wrap(mod, name, label, only_inside=True):
    fn = getattr(mod, name)
    def w(*a, **k):
        if only_inside and not inside[0]:
            return fn(*a, **k)
        t = time.perf_counter(); r = fn(*a, **k); T[label] += time.perf_counter() - t; C[label] += 1; return r
    setattr(mod, name, w)
wrap(torch, "empty_like", "empty_like")
wrap(torch, "_foreach_copy_", "_foreach_copy_")
orig = FusedMappingLoop._txn_begin
def tb(self):
    if self._txn is not None or self._replaying:
        return orig(self)
    inside[0] = True
    t = time.perf_counter(); torch.cuda.synchronize(); T["sync before (GPU still busy?)"] += time.perf_counter() - t
    t = time.perf_counter(); r = orig(self); T["_txn_begin total"] += time.perf_counter() - t; C["_txn_begin total"] += 1
    inside[0] = False
    return r
FusedMappingLoop._txn_begin = tb
for i, f in enumerate(frames):
    if i == n // 2: T.clear(); C.clear()
    sess.process(*f)
print({k: (round(1e3 * v / max(1, C["_txn_begin total"]), 3), C[k]) for k, v in T.items()})
