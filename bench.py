#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: mapping frames/sec (+ render ms) at 640x480 / 300k Gaussians.

One STEP = one iteration of the keyframe mapping loop (/root/reference/src/mapper.py:414-568) on the synthetic room
(SURVEY.md 8d): 10 window views + 2 random views rendered forward AND backward through the HIP rasterizer, the
mapping loss per view, the isotropy regulariser, one Adam step on all Gaussian parameters (6 groups), the xyz lr
update, and the exposure (keyframe) Adam step.  A mapped keyframe costs 60 such iterations + 1 prune pass
(configs/splat_slam.yaml:44, mapper.py:1113-1114), so   mapping frames/sec = steps/sec / 61.

N > 1 (weak scaling): every rank renders its own 12 views of a replicated map; gradients are summed with RCCL and the
Adam step runs on the summed gradient -- a 12*N-view batch per step.  value = N * steps/sec / 61 (keyframe-equivalents/s).

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

KINDS = ["preprocess_fwd", "tile_scan", "scatter", "unused3", "unused4", "blend_fwd_incl_tile_sort", "unused6", "blend_bwd", "preprocess_bwd"]
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3


def trace(msg):
    if os.environ.get("SPLAT_BENCH_TRACE"):
        print("[trace]", msg, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gaussians", type=int, default=300000)
    ap.add_argument("--camera", default="metric", choices=["metric", "replica", "tiny"])
    ap.add_argument("--views", type=int, default=16, help="keyframes in the map (10 window + pool of random views)")
    ap.add_argument("--loop", default="fused", choices=["fused", "autograd"],
                    help="fused: autograd-free C-ABI sequence (product path); autograd: torch.autograd mirror of the reference loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--refine-iters", type=int, default=200, help="final_refine iterations timed for refine it/s (0 = skip)")
    ap.add_argument("--profile-all", action="store_true", help="event-time every kernel kind (adds overhead)")
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback in the product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from splat_slam_amd import _native as nat
    from splat_slam_amd import synthetic as syn
    from splat_slam_amd.mapper import MappingLoop
    from splat_slam_amd.parallel import GradientSync
    import diff_gaussian_rasterization as dgr

    lib = nat.lib()
    torch.manual_seed(43)
    import numpy as np
    np.random.seed(43)

    intr = syn.INTRINSICS[args.camera]
    N = args.gaussians
    params = syn.room_parameters(N, seed=43, device=dev)
    # every rank maps a different set of views of the same room (weak scaling): rotate the orbit by rank
    cams = syn.make_views(params, args.views, intr, dev, seed=43 + rank)
    from splat_slam_amd.fused import FusedMappingLoop
    loop = (FusedMappingLoop if args.loop == "fused" else MappingLoop)(syn.DEFAULT_CONFIG, device=dev)
    loop.gaussians = syn.model_from_parameters(params, device=dev)
    loop.viewpoints = {c.uid: c for c in cams}
    loop.current_window = list(range(min(10, args.views)))
    loop.build_keyframe_optimizers()
    # keep the workload stationary: densify_and_prune fires when iteration_count % 150 == 50 (mapper.py:531-541) and
    # would change N (the metric is quoted AT 300k Gaussians); start right after such a point -> 149 clean iterations
    loop.iteration_count = 50
    if world > 1:
        if args.loop == "fused":
            loop.world = world                                  # flat accumulator buffer, one all-reduce per step
        else:
            loop.grad_sync = GradientSync(loop.gaussians, world)

    def steps(k):
        # exactly k iterations of the mapping loop, driven the way the reference drives it: map(window, iters=...) calls
        # (mapper.py:1113 uses iters=60; here up to 90 per call so that any --steps stays clear of the densification
        # points, see above); per-call bookkeeping is paid once per call, as in the reference
        while k > 0:
            n = min(k, 90)
            loop.iteration_count = 50
            loop.map(loop.current_window, iters=n)
            k -= n

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    trace("setup done")
    if args.warmup:
        steps(args.warmup)
    barrier()
    trace("warmup done")
    mask = (1 << 7) if not args.profile_all else (1 << len(KINDS)) - 1
    lib.sgr_profile_enable(mask)            # HIP events around blend_bwd only (12 pairs per step) on the launch stream
    t0 = time.perf_counter()
    steps(args.steps)
    host_issue = time.perf_counter() - t0          # time the host needed to ENQUEUE the steps (no sync inside)
    barrier()
    elapsed = time.perf_counter() - t0
    trace("timed loop done")
    ms = (C.c_float * len(KINDS))()
    cnt = (C.c_int64 * len(KINDS))()
    lib.sgr_profile_read(ms, cnt)
    lib.sgr_profile_enable(0)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = 1e3 * elapsed / args.steps
    views_per_step = len(loop.current_window) + min(2, args.views - len(loop.current_window))
    value = world * (args.steps / elapsed) / 61.0

    # ---- single-render timings + work counters (outside the timed region)
    from splat_slam_amd.renderer import render
    from splat_slam_amd.mapper import PipelineParams
    bg = loop.background
    cam0 = cams[0]

    def timed(fn, reps=10):
        torch.cuda.synchronize()
        a = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - a) / reps

    def fwd_only():
        with torch.no_grad():
            render(cam0, loop.gaussians, PipelineParams(), bg)

    def fwd_bwd():
        pkg = render(cam0, loop.gaussians, PipelineParams(), bg)
        loss = loop.loss_fn(loop.config["mapping"], pkg["render"], pkg["depth"], cam0, pkg["opacity"])
        loss.backward()
        loop.gaussians.optimizer.zero_grad(set_to_none=True)

    trace("profile read")
    fwd_only(); torch.cuda.synchronize(); trace("fwd_only ok"); fwd_bwd(); torch.cuda.synchronize(); trace("fwd_bwd ok")
    render_fwd_ms = timed(fwd_only)
    trace("timed fwd ok")
    render_fwd_bwd_ms = timed(fwd_bwd)
    trace("timed fwd_bwd ok")
    render_fused_ms = None
    if args.loop == "fused":       # the loop's own forward-only render (persistent buffers, no host sync): keyframe selection
        loop.render_forward(cam0)
        render_fused_ms = timed(lambda: loop.render_forward(cam0), reps=50)

    # ---- work counters of the views of the last timed step (saved blocks of the fused loop; one sync each)
    stats = (C.c_int64 * 4)()
    HW = intr["H"] * intr["W"]
    assert loop.gaussians.get_xyz.shape[0] == N, "N changed during the benchmark"
    per_view = []
    if args.loop == "fused":
        for cam in loop.last_used:
            vb = loop._views[cam.uid]
            ws = nat.SgrWorkspace(vb.saved.data_ptr(), vb.saved.numel(), vb.scratch.data_ptr(), vb.scratch.numel(), loop._cap)
            nat.check(lib.sgr_query_stats(C.byref(ws), N, intr["H"], intr["W"], vb.radii.data_ptr(), stats,
                                          torch.cuda.current_stream(dev).cuda_stream), "sgr_query_stats")
            per_view.append([int(x) for x in stats])
        hist = (C.c_int64 * 8)()
        nat.check(lib.sgr_query_list_histogram(C.byref(ws), N, intr["H"], intr["W"], hist, torch.cuda.current_stream(dev).cuda_stream),
                  "sgr_query_list_histogram")
        list_hist = dict(zip(["0", "1-4", "5-8", "9-16", "17-32", "33-64", "65-256", ">256"], [int(x) for x in hist]))
    else:
        list_hist = None
        pkg = render(cam0, loop.gaussians, PipelineParams(), bg)
        fn = pkg["render"].grad_fn
        saved = fn.saved_tensors[-1]
        st = dgr._state(dev)
        ws = nat.SgrWorkspace(saved.data_ptr(), saved.numel(), st.scratch.data_ptr(), st.scratch.numel(), fn.capacity)
        nat.check(lib.sgr_query_stats(C.byref(ws), N, intr["H"], intr["W"], pkg["radii"].data_ptr(), stats,
                                      torch.cuda.current_stream(dev).cuda_stream), "sgr_query_stats")
        per_view.append([int(x) for x in stats])
    trace("stats ok")
    # ---- final-refinement throughput (mapper.py:656-708: ONE random view fwd+bwd + Adam on all N per iteration);
    # measured after the counters were read because it keeps optimising the map.  Outside the headline timed region.
    refine_its = None
    if args.loop == "fused" and args.refine_iters > 0:
        loop.final_refine(iters=5)
        torch.cuda.synchronize()
        a = time.perf_counter()
        loop.final_refine(iters=args.refine_iters)
        torch.cuda.synchronize()
        refine_its = args.refine_iters / (time.perf_counter() - a)
        trace("refine ok")
    nv = len(per_view)
    V = sum(p[0] for p in per_view) // nv
    R = sum(p[1] for p in per_view) // nv
    R_eff_sum = sum(p[2] for p in per_view)
    tiles_nonempty = sum(p[3] for p in per_view) // nv

    # ---- roofline of the dominant kernel (tile-blend backward).  One launch covers `views_in_launch` views (batched).
    bwd_launches = int(cnt[7])
    bwd_ms = float(ms[7]) / max(1, bwd_launches)
    views_in_launch = nv if args.loop == "fused" else 1
    r_eff_launch = R_eff_sum if args.loop == "fused" else per_view[0][2]
    alg_bytes = 84 * r_eff_launch + (24 * HW + 40 * N) * views_in_launch          # SURVEY.md 8d figure
    # what this design moves: 92 B per pair; per pixel 9 B in the fused loop (final_T, n_contrib, one code byte of
    # loss-gradient signs) or 24 B through the autograd API (float gradients)
    own_bytes = 92 * r_eff_launch + (9 if args.loop == "fused" else 24) * HW * views_in_launch
    achieved = alg_bytes / (bwd_ms * 1e-3) / 1e9 if bwd_ms > 0 else 0.0
    pair_evals = r_eff_launch * 64               # (pixel, splat) pairs the kernel evaluates
    valu_tflops = pair_evals * 60 / (bwd_ms * 1e-3) / 1e12 if bwd_ms > 0 else 0.0
    traffic = None
    try:                                         # PMC pass of this same command, committed under profiles/
        pm = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc_hbm_bytes.json")))
        if pm.get("workload") == [N, intr["W"], intr["H"], views_in_launch]:
            name = "sgr::blend_bwd_kernel<true>" if args.loop == "fused" else "sgr::blend_bwd_kernel<false>"
            traffic = pm["kernels"][name]["hbm_bytes_per_launch_corrected"]
    except Exception:
        pass
    roofline = {"kernel": "blend_bwd_kernel", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "avg_launch_ms": round(bwd_ms, 5), "launches": bwd_launches, "views_per_launch": views_in_launch,
                "algorithmic_bytes": alg_bytes, "own_formula_bytes": own_bytes,
                "pixel_splat_pairs_per_launch": pair_evals,
                "valu_frac_at_60flop_per_pair": round(valu_tflops / FP32_PEAK_TFLOPS, 4),
                "note": "splat-list blending is VALU-issue bound (DESIGN.md 3): HBM fraction is small by construction"}

    out = {
        "metric": "mapping frames/sec + render ms @640x480, 300k Gaussians; PSNR vs ref",
        "value": round(value, 4), "unit": "mapped keyframes/s (61 map() iterations each)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]-shaped: synthetic room (SURVEY 8d), %d Gaussians, %dx%d, %d views/step "
                               "(10 window + 2 random) fwd+bwd + loss + isotropy + Adam, 1xMI355X per rank"
                               % (N, intr["W"], intr["H"], views_per_step),
                   "gaussians": N, "width": intr["W"], "height": intr["H"], "views_per_step": views_per_step, "loop": args.loop,
                   "parallelism": "view-parallel x%d, RCCL grad all-reduce" % world if world > 1 else "single GPU"},
        "host_enqueue_ms_per_step": round(1e3 * host_issue / args.steps, 4),
        "render_ms": {"forward": round(render_fwd_ms, 4), "forward_backward_loss": round(render_fwd_bwd_ms, 4),
                      "forward_fused_loop": None if render_fused_ms is None else round(render_fused_ms, 4),
                      "note": "forward / forward_backward_loss: one view through the drop-in autograd API (one host sync per "
                              "forward, like upstream); forward_fused_loop: FusedMappingLoop.render_forward, no sync"},
        "map_iterations_per_s": round(args.steps / elapsed, 2),
        "refine_iterations_per_s": None if refine_its is None else round(refine_its, 1),
        "work_per_view": {"visible_gaussians": V, "tile_pairs_R": R, "tile_pairs_walked_R_eff": R_eff_sum // nv,
                          "nonempty_tiles": tiles_nonempty, "tiles_by_walked_list_length_last_view": list_hist},
        "roofline": roofline,
    }
    if args.profile_all:
        out["kernel_ms"] = {k: round(float(ms[i]) / max(1, int(cnt[i])), 5) for i, k in enumerate(KINDS)}

    # ---- CPU baseline: the oracle (a port, not the reference: the reference has no CPU path) on the host cores
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(loop, cam0, intr, views_per_step)

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def usable_cores():
    """Host threads we may really use: affinity mask clipped by the cgroup CPU quota (the GPU box exposes 256 cores
    behind a 16-CPU quota; asking torch for 256 threads there makes the oracle crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(loop, cam, intr, views_per_step):
    """Times oracle/raster_oracle.py (PyTorch CPU, fp32) forward+backward of ONE view of the same scene."""
    from oracle import raster_oracle as O
    import math
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    gm = loop.gaussians
    with torch.no_grad():
        inp = dict(means3D=gm.get_xyz.detach().cpu(), opacities=gm.get_opacity.detach().cpu(),
                   shs=gm.get_features.detach().cpu(), scales=gm.get_scaling.detach().cpu(),
                   rotations=gm.get_rotation.detach().cpu())
        gt_img = cam.original_image.cpu()
        gt_dep = cam.depth.cpu()[None]
    s = O.OracleSettings(intr["H"], intr["W"], math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3), 1.0,
                         cam.world_view_transform.cpu(), cam.full_proj_transform.cpu(), cam.projection_matrix.cpu(), 0,
                         cam.camera_center.cpu(), False, False)
    x = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
    m2d = torch.zeros_like(x["means3D"], requires_grad=True)
    t0 = time.perf_counter()
    col, radii, dep, opa, nt = O.rasterize(x["means3D"], m2d, x["opacities"], shs=x["shs"], scales=x["scales"],
                                           rotations=x["rotations"], theta=torch.zeros(3, requires_grad=True),
                                           rho=torch.zeros(3, requires_grad=True), settings=s)
    t1 = time.perf_counter()
    loss = 0.8 * (col - gt_img).abs().mean() + 0.2 * (dep - gt_dep).abs().mean()
    loss.backward()
    t2 = time.perf_counter()
    sec_view = t2 - t0
    return {"value": round(1.0 / (sec_view * views_per_step * 61.0), 6), "unit": "mapped keyframes/s (61 map() iterations each)",
            "cores": ncores, "kind": "port",
            "sample": "1 view (of the %d per step) forward+backward of the same %d-Gaussian scene through "
                      "oracle/raster_oracle.py (PyTorch CPU fp32, %d threads); Adam excluded" % (views_per_step, inp["means3D"].shape[0], ncores),
            "forward_s": round(t1 - t0, 3), "backward_s": round(t2 - t1, 3)}


if __name__ == "__main__":
    main()
