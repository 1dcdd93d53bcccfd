"""GPU box: where the HOST time of the drop-in (autograd) mapping iteration goes -- cProfile over a few map() iterations of
splat_slam_amd.mapper.MappingLoop at the bench configuration.   python scripts/profile_dropin.py [iters]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sys.argv = [sys.argv[0]] + ["--no-extras", "--no-cpu-baseline"]
args = bench.parse()
B = bench.Bench(args)
loop, cams = B.build("autograd", 0.0)
B.run_steps(loop, 3)
torch.cuda.synchronize()
iters = 5
t0 = time.perf_counter()
B.run_steps(loop, iters)
host = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print("ms per iteration: host enqueue %.3f, wall %.3f" % (1e3 * host / iters, 1e3 * total / iters))
pr = cProfile.Profile()
pr.enable()
B.run_steps(loop, iters)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
