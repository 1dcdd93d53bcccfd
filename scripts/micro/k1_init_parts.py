"""GPU box: K1 (preprocess_fwd) on the INITIALISATION workload -- one view, 19-35 k Gaussians that are all visible and cover tens to
thousands of bins -- under one SGR_DEBUG setting (sgr_common.h: bits 0-7 switch parts of the kernel off; results are wrong then).
    for d in 0 1 16 32 64 128; do SGR_DEBUG=$d python scripts/micro/k1_init_parts.py; done"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splat_slam_amd import synthetic as syn
from splat_slam_amd.fused import FusedMappingLoop
from splat_slam_amd.session import MappingSession
dev = torch.device("cuda:0")
intr = syn.INTRINSICS["metric"]
torch.manual_seed(43); np.random.seed(43)
frames = syn.keyframe_stream(2, intr, dev, n_world=400000, seed=43, sweep_deg=360.0 / 160)
dbg = os.environ.pop("SGR_DEBUG", "0")
# the map is built WITHOUT the debug bits (a clean initialisation up to iteration 600), the timed iterations run in a child setting:
# the library reads SGR_DEBUG once, so the clean part runs in this process only when dbg == 0; otherwise the state is what the bits leave
os.environ["SGR_DEBUG"] = dbg
loop = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev)
loop.init_itr_num = 300
sess = MappingSession(loop, intr)
sess.process(*frames[0])
torch.cuda.synchronize()
cam = loop.viewpoints[0]
lr = float(loop._xyz_group()["lr"])
lib = loop.lib
loop._run_span([cam], [], [], [lr] * 10, 0.0, "none", initialization=True)
torch.cuda.synchronize()
lib.sgr_profile_enable(0x7f)
loop._run_span([cam], [], [], [lr] * 40, 0.0, "none", initialization=True)
torch.cuda.synchronize()
ms, cnt = (C.c_float * 7)(), (C.c_int64 * 7)()
lib.sgr_profile_read(ms, cnt)
lib.sgr_profile_enable(0)
names = ["preprocess_fwd", "tile_scan", "scatter", "blend_fused", "blend_fwd", "blend_bwd", "preprocess_bwd_incl_optimiser"]
print(json.dumps({"SGR_DEBUG": int(dbg), "gaussians": int(loop.gaussians.get_xyz.shape[0]),
                  "avg_ms": {n: round(float(ms[i]) / max(1, int(cnt[i])), 4) for i, n in enumerate(names) if int(cnt[i])}}))
