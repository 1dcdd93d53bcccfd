#!/usr/bin/env python3
"""configs[4]'s question (SURVEY.md 8e "semantics caveat"): the reference's final_refine takes ONE random view per Adam step
(src/mapper.py:656-708); spread over G GPUs it becomes G views per step and iters / G steps.  Does the map end up as good?

Builds a map with a mapping session (synthetic room, default hyper-parameters), snapshots it, and refines the SAME snapshot
with views_per_step = 1, 2, 4, 8 for the same number of renders (a single process runs the G views of a step as one batched
launch: the same optimisation as G ranks up to fp32 summation order, tests/test_gpu_round3.py).  Reports the PSNR over all
keyframes (eval_utils.py:90-123 protocol) and the wall time of each variant on one GPU.

    python scripts/refine_batch_psnr.py [--keyframes 100] [--refine 2000] [--out profiles/r03_refine_batch_psnr.json]"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keyframes", type=int, default=100)
    ap.add_argument("--refine", type=int, default=2000)
    ap.add_argument("--camera", default="metric")
    ap.add_argument("--groups", default="1,2,4,8")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from splat_slam_amd import synthetic as syn
    from splat_slam_amd.eval import eval_rendering_psnr
    from splat_slam_amd.fused import FusedMappingLoop
    from splat_slam_amd.mapper import PipelineParams
    from splat_slam_amd.session import MappingSession
    dev = torch.device("cuda:0")
    torch.manual_seed(43)
    np.random.seed(43)
    intr = syn.INTRINSICS[a.camera]
    frames = syn.keyframe_stream(a.keyframes, intr, dev, n_world=400000, seed=43, sweep_deg=360.0 * (a.keyframes - 1) / a.keyframes)
    loop = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev)
    sess = MappingSession(loop, intr)
    for f in frames:
        sess.process(*f)
    torch.cuda.synchronize()
    gm = loop.gaussians
    names = ["_xyz", "_features_dc", "_opacity", "_scaling", "_rotation"]
    groups = {g["name"]: g for g in gm.optimizer.param_groups}
    snap = {"p": {n: getattr(gm, n).detach().clone() for n in names},
            "opt": {k: {"m": gm.optimizer.state[g["params"][0]]["exp_avg"].clone(), "v": gm.optimizer.state[g["params"][0]]["exp_avg_sq"].clone(),
                        "step": float(gm.optimizer.state[g["params"][0]]["step"])} for k, g in groups.items() if g["params"][0].numel()},
            "exp": {k: getattr(loop._exp, k).clone() for k in ("param", "m", "v", "step")}, "it": loop.iteration_count,
            "lr": {k: g["lr"] for k, g in groups.items()}}
    cams = [loop.viewpoints[k] for k in sorted(loop.viewpoints)]

    def restore():
        with torch.no_grad():
            for n in names:
                getattr(gm, n).data.copy_(snap["p"][n])
            for k, s in snap["opt"].items():
                st = gm.optimizer.state[groups[k]["params"][0]]
                st["exp_avg"].copy_(s["m"])
                st["exp_avg_sq"].copy_(s["v"])
                st["step"].fill_(s["step"])
            for k, v in snap["exp"].items():
                getattr(loop._exp, k).copy_(v)
        for k, g in groups.items():
            g["lr"] = snap["lr"][k]
        loop.iteration_count = snap["it"]
        loop._plan_key = None                      # (the cached Adam step counters)
        gm.invalidate_activations()

    def psnr():
        with torch.no_grad():
            return eval_rendering_psnr(cams, gm, PipelineParams(), loop.background)

    rows = []
    base = psnr()
    rows.append({"views_per_step": 0, "optimiser_steps": 0, "renders": 0, "psnr_mean": round(float(np.mean(base)), 3), "psnr_min": round(float(np.min(base)), 3)})
    for G in [int(x) for x in a.groups.split(",")]:
        restore()
        np.random.seed(43)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop.final_refine(iters=a.refine, views_per_step=G)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sc = psnr()
        rows.append({"views_per_step": G, "optimiser_steps": -(-a.refine // G), "renders": a.refine, "psnr_mean": round(float(np.mean(sc)), 3),
                     "psnr_min": round(float(np.min(sc)), 3), "wall_s_one_gpu": round(dt, 3), "renders_per_s_one_gpu": round(a.refine / dt, 1)})
        print(json.dumps(rows[-1]), flush=True)
    out = {"what": "final_refine of the SAME snapshot with 1 vs G views per optimiser step, equal number of renders (mapper.py:656-708; configs[4])",
           "map": {"keyframes": len(cams), "gaussians": int(gm.get_xyz.shape[0]), "image": [intr["W"], intr["H"]]}, "rows": rows}
    print(json.dumps(out))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
