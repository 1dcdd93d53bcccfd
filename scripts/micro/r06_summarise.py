"""Copies what gpurun_out/<tag>/ holds (scripts/micro/r06_collect.sh) into profiles/ and prints the numbers DESIGN.md / BASELINE.md quote.
    python scripts/micro/r06_summarise.py [tag=r06]"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
for f in ("bench_kernel_stats.csv", "bench_line.json", "kernel_batched_avg.json", "pmc_hbm_bytes.json", "pmc_sq.json", "session_configs1.json",
          "multi_path_time_light.json", "bench_1500k.json", "table_numbers.json", "dropin_phases.json"):
    shutil.copy(os.path.join(src, tag + "_" + f), os.path.join(dst, tag + "_" + f))
for f in ("latest_kernel_batched_avg.json", "latest_pmc_hbm_bytes.json"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, f))
out = {"what": "scripts/micro/pmc_tile.py on the opaque scene (--scale-add 1.6) and the light scene, whole kernel and by region (SGR_DEBUG 4096: no backward; "
               "2048: no walk, no backward), final round-6 build; per wave = counter / SQ_WAVES of the fused tile kernel blend_fwd_kernel<512, true>",
       "round5_for_comparison": {"opaque": {"SQ_INSTS_VALU_per_wave": 5049, "backward": 3219, "walk": 1358, "rest": 473}, "light": {"SQ_INSTS_VALU_per_wave": 965}}}
for scene in ("opaque", "light"):
    r = {}
    for t in ("full", "nobwd", "nowalk"):
        d = json.load(open(os.path.join(src, "pmc_tile_%s_%s.json" % (scene, t))))
        r[t] = {"SGR_DEBUG": d["SGR_DEBUG"], "command": d["command"], "kernels": d["kernels"]}
    f = lambda t: r[t]["kernels"]["sgr::blend_fwd_kernel<512, true>"]["per_wave"]["SQ_INSTS_VALU"]
    r["valu_instructions_per_wave_by_region"] = {"whole_kernel": f("full"), "backward": round(f("full") - f("nobwd"), 1),
                                                 "forward_walk": round(f("nobwd") - f("nowalk"), 1), "everything_else": f("nowalk")}
    out[scene] = r
    print(scene, r["valu_instructions_per_wave_by_region"])
json.dump(out, open(os.path.join(dst, tag + "_pmc_tile_opaque.json"), "w"), indent=1)
k = json.load(open(os.path.join(dst, tag + "_kernel_batched_avg.json")))
print({n.replace("sgr::", ""): v["avg_us"] for n, v in k["kernels"].items() if v["views_per_launch"] > 1})
h = json.load(open(os.path.join(dst, tag + "_pmc_hbm_bytes.json")))["kernels"]["sgr::preprocess_fwd_kernel"]
print("K1 fetch/write KB", h["FETCH_SIZE_avg_KB_per_launch"], h["WRITE_SIZE_avg_KB_per_launch"])
sq = json.load(open(os.path.join(dst, tag + "_pmc_sq.json")))["kernels"]["sgr::preprocess_fwd_kernel"]
print("K1 wait frac", sq["frac_wait_memory_or_barrier"])
b = json.loads(open(os.path.join(dst, tag + "_bench_line.json")).read())
print("value", b["value"], "ms", b["ms_per_step"], "before", b["iterations_before_timed_region"])
for key in ("roofline", "roofline_opaque", "roofline_unfused_blend_bwd"):
    r = b[key]
    print(key, {q: r.get(q) for q in ("bound", "achieved", "frac", "hbm_frac_survey_formula", "hbm_frac_measured_traffic", "avg_launch_ms",
                                      "valu_instructions_per_wave", "valu_busy_share_of_launch", "traffic", "algorithmic_bytes")})
e = b["extra"]
print("opaque", e["opaque_scene"]["ms_per_step"], e["opaque_scene"]["kernel_ms"])
print("session", e["session"]["ms_per_keyframe"], e["session"]["ms_init_keyframe_1050_iterations"], e["session"]["psnr_all_keyframes_mean"], e["session"]["hip_vs_oracle_one_view"])
f = e["session_full"]
print("full", {q: f.get(q) for q in ("gaussians_final", "ms_per_mapped_keyframe", "keyframes_per_s", "ms_per_keyframe_by_quarter", "psnr_all_keyframes_mean",
                                     "hip_vs_oracle_one_view", "overflow_events")})
print("dropin", b["dropin"]["ms_per_iteration"], b["dropin"]["reference_getters_ms_per_iteration"], "refine", b.get("refine_iterations_per_s"), "render", b["render_ms"])
print("cpu", b["cpu_baseline"]["value"])
s = json.load(open(os.path.join(dst, tag + "_session_configs1.json")))
print("script session", {q: s[q] for q in ("gaussians_final", "ms_per_mapped_keyframe", "keyframes_per_s_incl_surgery", "psnr_all_keyframes_mean", "final_refine")})
m = [json.loads(l) for l in open(os.path.join(src, "multi.log")) if l.startswith("{")]
print("multi", [(x["world"], x["ms_per_iteration_rank0_no_collective_time"]) for x in m])
b15 = json.loads(open(os.path.join(dst, tag + "_bench_1500k.json")).read().strip().splitlines()[-1])
print("1.5M", b15["value"], b15["ms_per_step"])
