"""one row of numbers per configuration (run on the GPU box).
    python scripts/table_numbers.py  ->  gpurun_out/table_numbers.json"""
import json, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

CONFIGS = [("C0 room 640x320 N=20k", 20000, "replica"), ("C1 room 640x480 N=300k", 300000, "metric"),
           ("C1 room 640x320 N=300k", 300000, "replica")]


def psnr_vs_oracle(n, camera):
    from splat_slam_amd import synthetic as syn
    from splat_slam_amd.mapper import MappingLoop, PipelineParams
    from splat_slam_amd.renderer import render
    from oracle import raster_oracle as O
    dev = torch.device("cuda:0")
    intr = syn.INTRINSICS[camera]
    params = syn.room_parameters(n, seed=43, device=dev)
    cam = syn.make_views(params, 2, intr, dev, seed=43)[0]
    loop = MappingLoop(syn.DEFAULT_CONFIG, device=dev)
    gm = syn.model_from_parameters(params, device=dev)
    with torch.no_grad():
        pkg = render(cam, gm, PipelineParams(), loop.background)
        s = O.OracleSettings(intr["H"], intr["W"], math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3), 1.0,
                             cam.world_view_transform.cpu(), cam.full_proj_transform.cpu(), cam.projection_matrix.cpu(), 0,
                             cam.camera_center.cpu(), False, False)
        col, radii, dep, opa, nt = O.rasterize(gm.get_xyz.cpu(), torch.zeros(n, 3), gm.get_opacity.cpu(), shs=gm.get_features.cpu(),
                                               scales=gm.get_scaling.cpu(), rotations=gm.get_rotation.cpu(), settings=s)
    mse = float(((pkg["render"].cpu() - col) ** 2).mean())
    return {"psnr_hip_vs_oracle_db": round(10 * math.log10(1.0 / max(mse, 1e-20)), 1), "max_abs_diff": float((pkg["render"].cpu() - col).abs().max()),
            "radii_equal": bool(torch.equal(pkg["radii"].cpu(), radii.to(torch.int32)))}


rows = []
for name, n, cam in CONFIGS:
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gaussians", str(n), "--camera", cam, "--no-extras"]
    if n > 20000:
        args.append("--no-cpu-baseline")
    out = subprocess.run(args, capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    row = {"config": name, "fwd_ms": d["render_ms"]["forward"], "fwd_bwd_loss_ms": d["render_ms"]["forward_backward_loss"],
           "map_it_per_s": d["map_iterations_per_s"], "kf_per_s": d["value"], "refine_it_per_s": d["refine_iterations_per_s"],
           "hbm_frac_blend_bwd": d["roofline_unfused_blend_bwd"]["frac"],
           "valu_frac_blend_bwd": d["roofline_unfused_blend_bwd"]["valu_frac_at_60flop_per_pair"],
           "blend_bwd_ms": d["roofline_unfused_blend_bwd"]["avg_launch_ms"], "fused_tile_kernel_ms": d["roofline"]["avg_launch_ms"],
           "hbm_frac_fused_tile_kernel": d["roofline"]["frac"],
           "ms_per_step": d["ms_per_step"], "work_per_view": d["work_per_view"]}
    if "cpu_baseline" in d:
        row["cpu_oracle"] = d["cpu_baseline"]
    row.update(psnr_vs_oracle(n, cam))
    rows.append(row)
    print(json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "table_numbers.json"), "w"), indent=1)
