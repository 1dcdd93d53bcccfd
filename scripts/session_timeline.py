"""GPU box: host timeline of a mapped keyframe of the bench session (bench.py extra.session: 40 tracker frames, default
hyper-parameters): when the keyframe-selection decision is read back the GPU is idle until the first span of map() is enqueued --
how long that is, and what the host does meanwhile.   python scripts/session_timeline.py [--frames 40] [--out x.json]"""
import argparse
import collections
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--out", default=None)
a = ap.parse_args()
from splat_slam_amd import synthetic as syn  # noqa: E402
from splat_slam_amd.fused import FusedMappingLoop  # noqa: E402
from splat_slam_amd.session import MappingSession  # noqa: E402

dev = torch.device("cuda:0")
intr = syn.INTRINSICS["metric"]
torch.manual_seed(43)
np.random.seed(43)
frames = syn.keyframe_stream(a.frames, intr, dev, n_world=400000, seed=43, sweep_deg=360.0 * (a.frames - 1) / 160)
wl = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev)
ws = MappingSession(wl, intr)
for f in frames[:12]:
    ws.process(*f)
torch.cuda.synchronize()
del wl, ws
torch.manual_seed(43)
np.random.seed(43)
loop = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev)
sess = MappingSession(loop, intr)
marks = []


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def inner(*args, **kw):
        t0 = time.perf_counter()
        r = fn(*args, **kw)
        marks.append((label, t0, time.perf_counter()))
        return r
    setattr(obj, name, inner)


wrap(loop, "render_forward", "render_forward (+ header read)")
wrap(sess, "is_keyframe", "is_keyframe (read-back: the decision)")
wrap(sess, "add_to_window", "add_to_window (read-back)")
wrap(loop, "add_next_kf", "add_next_kf (seeding)")
wrap(loop, "build_keyframe_optimizers", "build_keyframe_optimizers")
wrap(loop, "_ensure_state", "_ensure_state")
wrap(loop, "_prepare_span", "_prepare_span")
wrap(loop, "_launch_span", "_launch_span (enqueue)")
wrap(loop, "_txn_commit", "_txn_commit (sync: capacity check)")
wrap(loop, "_txn_begin", "_txn_begin (snapshot)")
wrap(loop, "map", "map()")
rows = []
for i, f in enumerate(frames):
    del marks[:]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = sess.process(*f)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    if st != "mapped" or i < a.frames // 2:
        continue
    agg = collections.OrderedDict()
    for label, b, e in marks:
        agg[label] = agg.get(label, 0.0) + (e - b)
    dec = next((e for label, b, e in marks if label.startswith("is_keyframe")), None)
    first = next((b for label, b, e in marks if label.startswith("_launch_span")), None)
    rows.append({"wall_ms": 1e3 * t_all, "host_ms": 1e3 * t_host, "gpu_idle_decision_to_first_span_ms": 1e3 * (first - dec) if dec and first else None,
                 **{k: 1e3 * v for k, v in agg.items()}})
keys = list(rows[0].keys())
mean = {k: round(float(np.mean([r[k] for r in rows if r.get(k) is not None])), 3) for k in keys}
out = {"what": "mean over the mapped keyframes of the second half of the %d-frame bench session (ms per keyframe; nested entries overlap: map() "
               "contains the spans, the commits and _ensure_state)" % a.frames, "keyframes": len(rows), "gaussians": int(loop.gaussians.get_xyz.shape[0]),
       "mean_ms": mean}
print(json.dumps(out, indent=1))
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
