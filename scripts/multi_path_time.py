"""Host-side cost of the multi-GPU iteration path on ONE GPU: the all-reduce is replaced by an in-place self-add so that
the two C-ABI calls + collective slot per iteration are exercised without a second device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splat_slam_amd import synthetic as syn
from splat_slam_amd.fused import FusedMappingLoop
dev = torch.device("cuda:0")
torch.manual_seed(43); np.random.seed(43)
intr = syn.INTRINSICS["metric"]
params = syn.room_parameters(300000, seed=43, device=dev)
cams = syn.make_views(params, 16, intr, dev, seed=43)
for world, span in ((1, True), (2, True), (2, False)):
    loop = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev, span_calls=span)
    loop.gaussians = syn.model_from_parameters(params, device=dev)
    loop.viewpoints = {c.uid: c for c in cams}
    loop.current_window = list(range(10))
    loop.build_keyframe_optimizers()
    loop.iteration_count = 50
    loop.world = world
    loop._all_reduce_sum = lambda t: t.mul_(1.0)          # one tiny-kernel stand-in for the collective
    loop.map(loop.current_window, iters=10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.map(loop.current_window, iters=60)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("world", world, "span", span, "ms/step %.4f" % (1e3 * tot / 60), "host enqueue ms/step %.4f" % (1e3 * host / 60), flush=True)
