"""The JSON line bench.py printed on the MI355X (committed under profiles/) carries every key of the driver's contract."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line.json")))
    assert files, "no committed bench line under profiles/"
    return json.loads(open(files[-1]).read().strip().splitlines()[-1])


def test_bench_line_has_the_contract_keys():
    d = _latest_line()
    for k in ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"]:
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == ("none" if d["n_gpus"] == 1 else d["scaling_when_sharded"]) and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f32" and d["n_gpus"] == 1 and "workload" in d["config"] and "model" not in d["config"]
    assert "640x480" in d["metric"] and d["config"]["gaussians"] == 300000 and d["config"]["width"] == 640 and d["config"]["height"] == 480
    r = d["roofline"]
    for k in ["bound", "achieved", "peak", "unit", "frac", "traffic"]:
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma", "valu") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["unit"] == {"hbm": "GB/s", "mfma": "TFLOP/s", "valu": "G wave-instructions/s"}[r["bound"]]
    assert r["traffic"] is None or r["traffic"] > 0
    # round 6: the tile kernel is VALU-issue bound and the line says so -- `frac` is the share of the chip's VALU
    # issue slots (SQ_INSTS_VALU child pass of the same run); the two HBM readings rounds 1-5 mixed up carry their own names
    assert r["bound"] == "valu", "the PMC child passes of the committed line did not run"
    for k in ["hbm_frac_survey_formula", "hbm_frac_measured_traffic", "valu_instructions_per_wave", "waves_per_launch", "valu_busy_share_of_launch"]:
        assert k in r, k
    assert 0.0 < r["hbm_frac_measured_traffic"] < r["hbm_frac_survey_formula"] < 1.0 and 0.0 < r["frac"] <= 1.0
    assert abs(r["hbm_frac_measured_traffic"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 8e12) < 2e-4
    assert abs(r["achieved"] - r["valu_instructions_per_wave"] * r["waves_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    assert d["settle_iterations_untimed"] == 0 and d["iterations_before_timed_region"] >= d["warmup"]
    full = d["extra"]["session_full"]
    for k in ["gaussians_final", "ms_per_mapped_keyframe", "keyframes_per_s", "psnr_all_keyframes_mean", "hip_vs_oracle_one_view"]:
        assert k in full, k
    assert full["frames"] == 160 and full["keyframes_mapped"] >= 100 and full["gaussians_final"] > 150000
    assert abs(full["hip_vs_oracle_one_view"]["psnr_hip_render"] - full["hip_vs_oracle_one_view"]["psnr_oracle_render"]) < 0.01
    c = d["cpu_baseline"]
    for k in ["value", "unit", "cores", "kind", "sample"]:
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    # value is whole-job keyframes/s: consistent with the step time it was derived from (61 iterations per keyframe)
    assert abs(d["value"] - d["n_gpus"] * 1000.0 / d["ms_per_step"] / 61.0) / d["value"] < 0.01


def test_committed_profiles_agree_with_the_bench_line():
    """`roofline` describes the dominant kernel of the timed loop (round 4 on: the fused tile kernel; earlier lines: blend_bwd); its
    event-timed launch duration agrees with the committed rocprofv3 trace of the same command, its traffic with the PMC pass."""
    d = _latest_line()
    tag = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line.json")))[-1].split(os.sep)[-1].split("_")[0]
    k = json.load(open(os.path.join(ROOT, "profiles", tag + "_kernel_batched_avg.json")))["kernels"]
    hb = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_hbm_bytes.json")))["kernels"]
    legs = [(d["roofline"], "sgr::blend_fwd_kernel<512, true>" if "FUSED" in d["roofline"]["kernel"] else "sgr::blend_bwd_kernel<true>")]
    if "roofline_unfused_blend_bwd" in d:
        assert d["roofline"]["in_timed_region"] is True
        legs.append((d["roofline_unfused_blend_bwd"], "sgr::blend_bwd_kernel<true>"))
    for r, name in legs:
        trace_us = k[name]["avg_us"]
        event_us = 1e3 * r["avg_launch_ms"]
        assert abs(trace_us - event_us) / event_us < 0.10, (name, trace_us, event_us)      # rocprofv3 trace vs live HIP events
        h = hb[name]
        assert r["traffic"] in (None, h["hbm_bytes_per_launch_corrected"]) or \
            abs(r["traffic"] - h["hbm_bytes_per_launch_corrected"]) / h["hbm_bytes_per_launch_corrected"] < 0.05
