"""GPU box: allocator misses (hipMalloc calls of torch's caching allocator: memory_stats num_device_alloc) and host time per phase of a
mapped keyframe of the bench session -- a map that grows at every keyframe asks for sizes the cache has never seen.
    python scripts/micro/alloc_misses.py [frames=40]"""
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
T, A, C = collections.Counter(), collections.Counter(), collections.Counter()
na = lambda: torch.cuda.memory_stats().get("num_device_alloc", 0)
def wrap(obj, name, label=None):
    fn = getattr(obj, name); label = label or name
    def w(*a, **k):
        t, m = time.perf_counter(), na(); r = fn(*a, **k); T[label] += time.perf_counter() - t; A[label] += na() - m; C[label] += 1; return r
    setattr(obj, name, w)
for nm in ("render_forward", "add_next_kf", "build_keyframe_optimizers", "map", "_txn_begin", "_ensure_state", "_prepare_span", "_launch_span", "_txn_commit"):
    wrap(loop, nm)
for nm in ("is_keyframe", "add_to_window", "_camera"):
    wrap(sess, nm)
gm_cls = type(loop.gaussians) if loop.gaussians is not None else None
from splat_slam_amd import gaussian_model as gmod
for nm in ("densify_and_prune", "prune_points", "densification_postfix", "extend_from_pcd_seq", "reset_opacity", "reset_opacity_nonvisible", "cat_tensors_to_optimizer", "_prune_optimizer", "create_pcd_from_image_and_depth"):
    if hasattr(gmod.GaussianModel, nm): wrap(gmod.GaussianModel, nm, "gm." + nm)
kf = 0
for i, f in enumerate(frames):
    if i == n // 2: T.clear(); A.clear(); C.clear(); kf = 0; t0 = time.perf_counter(); torch.cuda.synchronize()
    r = sess.process(*f)
    kf += r == "mapped"
torch.cuda.synchronize()
print("mapped keyframes", kf, "ms per keyframe", round(1e3 * (time.perf_counter() - t0) / kf, 2))
for k in sorted(T, key=lambda k: -T[k]):
    print("%-36s %7.3f ms/kf  %6.1f device allocs/kf  %5.1f calls/kf" % (k, 1e3 * T[k] / kf, A[k] / kf, C[k] / kf))
