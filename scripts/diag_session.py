import copy, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from splat_slam_amd import synthetic as syn
from splat_slam_amd.fused import FusedMappingLoop
from splat_slam_amd.session import MappingSession
DEV = "cuda:0"
cfg = copy.deepcopy(syn.DEFAULT_CONFIG)
tr = cfg["mapping"]["Training"]
tr["init_itr_num"], tr["mapping_itr_num"], tr["window_size"] = 120, 12, 4
tr["init_gaussian_update"], tr["init_gaussian_reset"] = 40, 10 ** 9
cfg["mapping"]["opt_params"]["densify_from_iter"] = 10 ** 9
intr = syn.INTRINSICS["tiny"]
frames = syn.keyframe_stream(7, intr, DEV, n_world=20000, seed=5, sweep_deg=70.0)
torch.manual_seed(43); np.random.seed(43)
loop = FusedMappingLoop(cfg, device=DEV)
if len(sys.argv) > 1: loop.span_calls = bool(int(sys.argv[1]))
if len(sys.argv) > 2: loop.fuse_tail = bool(int(sys.argv[2]))
sess = MappingSession(loop, intr)
def report(tag):
    gm = loop.gaussians
    bad = {n: int((~torch.isfinite(getattr(gm, n))).sum()) for n in ["_xyz", "_features_dc", "_opacity", "_scaling", "_rotation"]}
    ex = None if loop._exp is None else loop._exp.param[:8].flatten().tolist()
    print(tag, "N", gm._xyz.shape[0], "nonfinite", bad, "exp", ex, flush=True)
for k, f in enumerate(frames):
    st = sess.process(*f)
    report("frame %d %s" % (k, st))
loop.final_refine(iters=40)
report("refined")
