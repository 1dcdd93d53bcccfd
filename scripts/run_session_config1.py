#!/usr/bin/env python3
"""configs[1] as SURVEY.md 8d restates it: a whole mapping SESSION -- keyframe after keyframe through MappingSession with the
reference's default hyper-parameters (configs/splat_slam.yaml: 1050 init iterations, 60 + 1 iterations per keyframe,
densify every 150 iterations, opacity resets, 1/32 of the pixels seeded per keyframe), map deformation of moved keyframes,
final refinement -- on the synthetic room (no dataset / tracker on the GPU box: the feed is splat_slam_amd.synthetic).

Reports what bench.py's stationary headline leaves out: wall-clock keyframes/s INCLUDING seeding (knn), densify / prune,
deformation and keyframe management; the amortised cost of map surgery; the map size it converges to; and metric item
(iv) of SURVEY.md 8d -- PSNR of the converged map rendered by the HIP rasterizer vs the SAME map rendered by the oracle
(eval protocol of /root/reference/src/utils/eval_utils.py:90-123: exposure affine for k > 0, clamp, pixels with gt > 0).

    python scripts/run_session_config1.py [--keyframes 160] [--camera metric] [--refine 2000] [--oracle-views 3] [--out x.json]
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def psnr(img, gt):
    mask = gt > 0                                             # eval_utils.py:109,123
    mse = ((img[mask] - gt[mask]) ** 2).mean()
    return float(20 * torch.log10(1.0 / torch.sqrt(mse)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keyframes", type=int, default=160)
    ap.add_argument("--camera", default="metric", choices=["metric", "replica", "tiny"])
    ap.add_argument("--refine", type=int, default=2000, help="final_refine iterations (Replica: 2000, replica.yaml:18)")
    ap.add_argument("--oracle-views", type=int, default=3)
    ap.add_argument("--world", type=int, default=400000, help="Gaussians of the ground-truth room the keyframes observe")
    ap.add_argument("--moved-every", type=int, default=10, help="every k-th past keyframe gets a refined pose / depth (0 = none)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--profile", action="store_true", help="HIP-event time of every kernel kind over the mapping phase (adds overhead)")
    ap.add_argument("--max-gaussians", type=int, default=4000000)
    ap.add_argument("--warmup-frames", type=int, default=24, help="frames of an untimed throw-away session first (a fresh process "
                    "pays code loading, allocator growth and cold caches once: 134 vs 110 ms per keyframe measured)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from splat_slam_amd import synthetic as syn
    from splat_slam_amd.fused import FusedMappingLoop
    from splat_slam_amd.session import MappingSession
    dev = torch.device("cuda:0")
    torch.manual_seed(43)
    np.random.seed(43)
    intr = syn.INTRINSICS[a.camera]
    t0 = time.perf_counter()
    frames = syn.keyframe_stream(a.keyframes, intr, dev, n_world=a.world, seed=43, sweep_deg=360.0 * (a.keyframes - 1) / a.keyframes)
    torch.cuda.synchronize()
    t_feed = time.perf_counter() - t0

    # the tracker occasionally refines past keyframes (mapper.py:1021-1055): a small rigid correction + a depth rescale
    moved = {}

    def pose_source(kf):
        if not a.moved_every or kf == 0 or kf % a.moved_every or moved.get(kf):
            return None
        moved[kf] = True
        _, _, _, depth, w2c = frames[kf]
        d = torch.eye(4)
        ang = 0.002
        d[:3, :3] = torch.tensor([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
        d[:3, 3] = torch.tensor([0.002, -0.001, 0.0015])
        return d @ w2c, depth * 1.002

    if a.warmup_frames > 0:
        wl = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev)
        ws = MappingSession(wl, intr)
        for f in frames[: a.warmup_frames]:
            ws.process(*f)
        torch.cuda.synchronize()
        del wl, ws
        torch.manual_seed(43)
        np.random.seed(43)
    loop = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev)
    sess = MappingSession(loop, intr, pose_source=pose_source)
    surgery_s = [0.0]
    gm_cls = type(loop.gaussians)
    for name in ("densify_and_prune", "extend_from_pcd_seq", "reset_opacity", "reset_opacity_nonvisible"):
        orig = getattr(gm_cls, name)

        def timed(self, *args, __orig=orig, **kw):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = __orig(self, *args, **kw)
            torch.cuda.synchronize()
            surgery_s[0] += time.perf_counter() - t
            return r
        setattr(gm_cls, name, timed)
    status, n_hist, t_kf = [], [], []
    if a.profile:
        loop.lib.sgr_profile_enable(0x7f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in frames:
        t1 = time.perf_counter()
        status.append(sess.process(*f))
        torch.cuda.synchronize()
        t_kf.append(time.perf_counter() - t1)
        n_hist.append(int(loop.gaussians.get_xyz.shape[0]))
        if a.verbose:
            worst = max((vb.pairs for vb in loop._views.values()), default=0)
            print("[kf %3d] %-7s N %8d  pairs(max over cameras) %9d  capacity %9d  %.2f s  mem %.1f GB" % (
                f[0], status[-1], n_hist[-1], worst, loop._cap, t_kf[-1], torch.cuda.memory_allocated() / 2 ** 30), file=sys.stderr, flush=True)
        if n_hist[-1] > a.max_gaussians:
            raise RuntimeError("map grew to %d Gaussians (> --max-gaussians): the synthetic feed does not converge" % n_hist[-1])
    t_map = time.perf_counter() - t0
    mapped = status.count("mapped")
    kernel_ms = None
    if a.profile:
        import ctypes as C
        ms, cnt = (C.c_float * 7)(), (C.c_int64 * 7)()
        loop.lib.sgr_profile_read(ms, cnt)
        loop.lib.sgr_profile_enable(0)
        names = ["preprocess_fwd", "tile_scan", "scatter", "blend_fused", "blend_fwd", "blend_bwd", "preprocess_bwd"]
        kernel_ms = {n: {"total_s": round(float(ms[i]) / 1e3, 3), "launches": int(cnt[i]), "avg_ms": round(float(ms[i]) / max(1, int(cnt[i])), 4)}
                     for i, n in enumerate(names) if int(cnt[i])}
        from splat_slam_amd import _native as nat
        cam = loop.last_used[0]
        vb = loop.workspace_of(cam)
        ws = nat.SgrWorkspace(vb.saved.data_ptr(), vb.saved.numel(), vb.scratch.data_ptr(), vb.scratch.numel(), loop._cap)
        hist = (C.c_int64 * 8)()
        nat.check(loop.lib.sgr_query_list_histogram(C.byref(ws), int(loop.gaussians.get_xyz.shape[0]), intr["H"], intr["W"], hist,
                                                    torch.cuda.current_stream().cuda_stream), "hist")
        kernel_ms["tiles_by_walked_list_length_last_view"] = dict(zip(["0", "1-4", "5-8", "9-16", "17-32", "33-64", "65-256", ">256"],
                                                                      [int(x) for x in hist]))
    t1 = time.perf_counter()
    scores = sess.finish(refine_iters=a.refine)
    torch.cuda.synchronize()
    t_refine = time.perf_counter() - t1
    N = int(loop.gaussians.get_xyz.shape[0])

    # ---- PSNR of the converged map: HIP render vs oracle render of the SAME map, a few keyframes
    from oracle import raster_oracle as O
    from splat_slam_amd.mapper import PipelineParams
    from splat_slam_amd.renderer import render
    gm = loop.gaussians
    ncores = max(1, min(len(os.sched_getaffinity(0)), 16))
    torch.set_num_threads(ncores)
    with torch.no_grad():
        inp = dict(means3D=gm.get_xyz.cpu(), opacities=gm.get_opacity.cpu(), shs=gm.get_features.cpu(), scales=gm.get_scaling.cpu(),
                   rotations=gm.get_rotation.cpu())
    keys = sorted(loop.viewpoints)
    pick = [keys[int(round(i * (len(keys) - 1) / max(1, a.oracle_views - 1)))] for i in range(a.oracle_views)]
    cmp_rows = []
    for k in pick:
        cam = loop.viewpoints[k]
        with torch.no_grad():
            img_h = render(cam, gm, PipelineParams(), loop.background)["render"]
            s = O.OracleSettings(intr["H"], intr["W"], math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3), 1.0,
                                 cam.world_view_transform.cpu(), cam.full_proj_transform.cpu(), cam.projection_matrix.cpu(), 0,
                                 cam.camera_center.cpu(), False, False)
            img_o = O.rasterize(inp["means3D"], None, inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                                rotations=inp["rotations"], settings=s)[0]
            ea, eb = (torch.exp(cam.exposure_a).item(), cam.exposure_b.item()) if k > 0 else (1.0, 0.0)      # eval_utils.py:96-99
            gt = cam.original_image.cpu()
            ph = psnr(torch.clamp(ea * img_h.cpu() + eb, 0.0, 1.0), gt)
            po = psnr(torch.clamp(ea * img_o + eb, 0.0, 1.0), gt)
            cmp_rows.append({"keyframe": int(k), "psnr_hip_map_hip_render": round(ph, 3), "psnr_hip_map_oracle_render": round(po, 3),
                             "image_max_abs_diff": float((img_h.cpu() - img_o).abs().max())})
    out = {
        "config": "configs[1]-shaped session: synthetic room, %dx%d, %d tracker frames, default splat_slam.yaml hyper-parameters"
                  % (intr["W"], intr["H"], len(frames)),
        "frames": len(frames), "keyframes_mapped": mapped, "skipped": status.count("skipped"),
        "gaussians_final": N, "gaussians_max": max(n_hist), "gaussians_after_init": n_hist[0],
        "wall_s_mapping_incl_init": round(t_map, 3), "wall_s_init_keyframe": round(t_kf[0], 3),
        "keyframes_per_s_incl_surgery": round(mapped / (t_map - t_kf[0]), 3),
        "ms_per_mapped_keyframe": round(1e3 * (t_map - t_kf[0]) / max(1, mapped), 3),
        "map_surgery_s_total": round(surgery_s[0], 3),
        "map_surgery_ms_per_keyframe": round(1e3 * surgery_s[0] / max(1, mapped + 1), 3),
        "final_refine": {"iters": a.refine, "wall_s": round(t_refine, 3), "it_per_s": round(a.refine / t_refine, 1) if a.refine else None},
        "warmup_frames_untimed": a.warmup_frames, "feed_s_rendering_ground_truth": round(t_feed, 2), "moved_keyframes_deformed": len(moved),
        "overflow_events": loop.overflow_events, "kernel_times_mapping_phase": kernel_ms,
        "psnr_all_keyframes_mean": round(float(np.mean(scores)), 3), "psnr_min": round(float(np.min(scores)), 3),
        "psnr_hip_vs_oracle_render_of_the_same_map": cmp_rows, "oracle_threads": ncores,
    }
    print(json.dumps(out))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
