"""GPU box: the tile kernels on the bench scenes -- blend_fwd and blend_bwd launched separately (SGR_OPT_FUSED_BLEND = 0) and
the fused tile kernel, HIP-event averages per 12-view launch, with the work counters of the scene.
    python scripts/tile_kernel_times.py [light opaque ...]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

which = sys.argv[1:] or ["light", "opaque"]
sys.argv = [sys.argv[0], "--no-extras", "--no-cpu-baseline"]
args = bench.parse()
B = bench.Bench(args)
SCENES = {"light": (0.0, None), "opaque": (1.6, None), "light_kf": (0.0, "keyframe"), "opaque_kf": (1.6, "keyframe"), "mid": (0.8, None)}
for name in which:
    scale_add, order = SCENES[name]
    loop, cams = B.build("fused", scale_add, order=order)
    B.run_steps(loop, 10)
    loop.check_overflow()
    el, _ = B.timed(loop, 40)
    per_view, hist = B.work_counters(loop)
    roof, roof_f = B.rooflines(loop, per_view, 20)
    nv = len(per_view)
    print(json.dumps({"scene": name, "ms_per_step": round(1e3 * el / 40, 4), "blend_fwd_ms": roof["blend_fwd_avg_launch_ms"],
                      "blend_bwd_ms": roof["avg_launch_ms"], "blend_bwd_frac": roof["frac"], "fused_ms": roof_f["avg_launch_ms"],
                      "fused_frac": roof_f["frac"], "pairs_per_view": sum(p[1] for p in per_view) // nv,
                      "walked_per_view": sum(p[2] for p in per_view) // nv, "longest_list": loop._max_list(), "hist": hist}), flush=True)
    del loop, cams
    torch.cuda.empty_cache()
