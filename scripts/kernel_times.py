"""GPU box: HIP-event time of every kernel kind of the mapping iteration on the bench scenes (light / opaque, random /
keyframe order), one line of JSON per scene.   python scripts/kernel_times.py [light opaque light_kf opaque_kf]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

which = sys.argv[1:] or ["light", "opaque", "light_kf", "opaque_kf"]
sys.argv = [sys.argv[0], "--no-extras", "--no-cpu-baseline"]
args = bench.parse()
B = bench.Bench(args)
SCENES = {"light": (0.0, None), "opaque": (1.6, None), "light_kf": (0.0, "keyframe"), "opaque_kf": (1.6, "keyframe")}
for name in which:
    scale_add, order = SCENES[name]
    loop, cams = B.build("fused", scale_add, order=order)
    B.run_steps(loop, 10)
    el, _ = B.timed(loop, 40)
    allk = B.profiled(loop, 20, (1 << len(bench.KINDS)) - 1, fused_blend=True)
    print(json.dumps({"scene": name, "ms_per_step": round(1e3 * el / 40, 4),
                      "kernel_ms": {bench.KINDS[i]: round(ms, 5) for i, (ms, n) in allk.items() if n}}), flush=True)
    del loop, cams
    torch.cuda.empty_cache()
