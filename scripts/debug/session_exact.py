"""GPU box: is a session whose capacity overflows (and is replayed) bit-identical to the same session with a capacity floor no view
reaches?  20 tracker frames, default hyper-parameters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splat_slam_amd import synthetic as syn
from splat_slam_amd.fused import FusedMappingLoop
from splat_slam_amd.session import MappingSession
dev = torch.device("cuda:0")
intr = syn.INTRINSICS["metric"]
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 20
def run(floor, verify):
    torch.manual_seed(43); np.random.seed(43)
    frames = syn.keyframe_stream(NF, intr, dev, n_world=400000, seed=43, sweep_deg=360.0 * (NF - 1) / 160)
    torch.manual_seed(43); np.random.seed(43)
    loop = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev)
    loop.capacity_floor = floor
    loop.verify_estimates = verify
    sess = MappingSession(loop, intr)
    st = [sess.process(*f) for f in frames]
    torch.cuda.synchronize()
    gm = loop.gaussians
    return {"N": gm._xyz.shape[0], "xyz": gm._xyz.detach().clone(), "op": gm._opacity.detach().clone(), "sc": gm._scaling.detach().clone(),
            "events": loop.overflow_events, "replays": loop.replayed_transactions, "status": st}
a = run(1 << 22, False)
b = run(1 << 16, False)
c = run(1 << 16, True)
for name, r in (("ample", a), ("default", b), ("default+verify", c)):
    print(name, "N", r["N"], "events", r["events"], "replays", r["replays"], r["status"].count("mapped"))
for name, r in (("default", b), ("default+verify", c)):
    same = r["N"] == a["N"] and all(torch.equal(r[k], a[k]) for k in ("xyz", "op", "sc"))
    print(name, "bitwise equal to ample:", same)
