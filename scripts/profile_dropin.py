"""GPU box: where the HOST time of the drop-in (autograd) mapping iteration goes.  Phase timers (time.perf_counter around the
Python entry points: no profiler overhead) over map() iterations of splat_slam_amd.mapper.MappingLoop at the bench
configuration, then a cProfile listing of the same loop.

    python scripts/profile_dropin.py [--iters 20] [--no-batch] [--out gpurun_out/dropin_phases.json]"""
import os
os.environ["SPLAT_RASTER_PROF"] = "1"
import argparse
import cProfile
import collections
import io
import json
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--no-batch", action="store_true")
ap.add_argument("--out", default=None)
a = ap.parse_args()
sys.argv = [sys.argv[0], "--no-extras", "--no-cpu-baseline"]
args = bench.parse()
B = bench.Bench(args)
import diff_gaussian_rasterization as drg  # noqa: E402
from splat_slam_amd import mapper as mapper_mod  # noqa: E402

if a.no_batch:
    drg.BATCH = False
loop, cams = B.build("autograd", 0.0)
B.run_steps(loop, 3)
torch.cuda.synchronize()
t0 = time.perf_counter()
B.run_steps(loop, a.iters)
host = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
res = {"batched_backward": drg.BATCH, "ms_per_iteration_host_enqueue": round(1e3 * host / a.iters, 3),
       "ms_per_iteration_wall": round(1e3 * total / a.iters, 3), "keyframes_per_s": round(a.iters / total / 61.0, 3)}

T = collections.Counter()
CNT = collections.Counter()
SAMPLES = collections.defaultdict(list)


def timed(name, fn):
    def w(*args, **kw):
        t = time.perf_counter()
        try:
            return fn(*args, **kw)
        finally:
            dt = time.perf_counter() - t
            T[name] += dt
            CNT[name] += 1
            SAMPLES[name].append(round(1e6 * dt, 1))
    return w


mapper_mod.render = timed("render() incl. rasterizer forward", mapper_mod.render)
ext = drg.native_extension()
res["native_nodes"] = ext is not None
drg.rasterize_gaussians = timed("  of which rasterize_gaussians (one call into _dgr.so, or the Python node)", drg.rasterize_gaussians)
if ext is not None:
    ext.profile_enable(True)
drg._RasterizeGaussians.forward = staticmethod(timed("  of which _RasterizeGaussians.forward body", drg._RasterizeGaussians.forward))
drg._RasterizeGaussians.backward = staticmethod(timed("  of which _RasterizeGaussians.backward (record)", drg._RasterizeGaussians.backward))
drg._batched_backward = timed("  of which collector: batched backward", drg._batched_backward)
loop.loss_fn = timed("loss_fn (fused mapping loss forward)", loop.loss_fn)
loop._visible_stats = timed("densification statistics", loop._visible_stats)
torch.Tensor.backward = timed("loss.backward() (all of autograd)", torch.Tensor.backward)
loop.gaussians.optimizer.step = timed("Gaussian optimizer.step", loop.gaussians.optimizer.step)
_bko = loop.build_keyframe_optimizers
if loop.keyframe_optimizers is not None:
    loop.keyframe_optimizers.step = timed("keyframe optimizer.step", loop.keyframe_optimizers.step)
torch.cuda.synchronize()
t0 = time.perf_counter()
B.run_steps(loop, a.iters)
wall = time.perf_counter() - t0
torch.cuda.synchronize()
# median per call x calls per iteration: one-off stalls (first use of a multi-tensor kernel, allocator growth) do not belong here
import statistics
res["phases_ms_per_iteration"] = {k: round(1e-3 * statistics.median(SAMPLES[k]) * CNT[k] / a.iters, 4)
                                  for k, v in sorted(T.items(), key=lambda kv: -statistics.median(SAMPLES[kv[0]]) * CNT[kv[0]])}
res["phases_ms_per_iteration_mean_incl_one_off_stalls"] = {k: round(1e3 * v / a.iters, 4) for k, v in T.items()}
res["calls_per_iteration"] = {k: round(v / a.iters, 2) for k, v in CNT.items()}
res["instrumented_ms_per_iteration_host"] = round(1e3 * wall / a.iters, 3)
res["collector_ms_per_iteration"] = ({k: round(1e3 * v / a.iters, 4) for k, v in ext.profile_read().items()} if ext is not None else
                                      {k: round(1e3 * v / (a.iters * 2 + 3), 4) for k, v in drg._PROF.items()})
res["first_calls_us"] = {k: v[:40] for k, v in SAMPLES.items()}
print(json.dumps({k: v for k, v in res.items() if k != "first_calls_us"}, indent=1))
print("first calls (us):", json.dumps(res["first_calls_us"]))
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
pr = cProfile.Profile()
pr.enable()
B.run_steps(loop, 5)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:7000])
