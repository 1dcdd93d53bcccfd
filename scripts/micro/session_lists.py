import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
from splat_slam_amd import synthetic as syn
from splat_slam_amd.fused import FusedMappingLoop
from splat_slam_amd.session import MappingSession
dev = torch.device("cuda:0"); intr = syn.INTRINSICS["metric"]
torch.manual_seed(43); np.random.seed(43)
frames = syn.keyframe_stream(40, intr, dev, n_world=400000, seed=43, sweep_deg=360.0 * 39 / 160)
for rep in range(2):
    torch.manual_seed(43); np.random.seed(43)
    loop = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev); sess = MappingSession(loop, intr)
    rows = []
    for f in frames:
        torch.cuda.synchronize(); t = time.perf_counter()
        st = sess.process(*f); torch.cuda.synchronize()
        rows.append((round(1e3 * (time.perf_counter() - t), 1), loop._max_list(), loop._build_class(), int(loop.gaussians.get_xyz.shape[0]), loop._cap, loop.replayed_transactions))
    print(rows)
