"""Runs on the GPU box: rocprofv3 kernel-trace + PMC passes of bench.py, reduced to small JSON/CSV summaries.

    python scripts/collect_profiles.py <tag>      ->  gpurun_out/<tag>/...

Per kernel the numbers are restricted to the BATCHED launches of the timed mapping loop (largest grid of that kernel):
bench.py also issues single-view launches (probe renders, render-ms timings) that would dilute a plain average.
PMC passes follow MI355X_MICROARCH.md: counters in their own runs (--pmc + --kernel-trace only), run from /tmp."""
import collections, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = os.path.join(ROOT, "gpurun_out", tag)
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
# the profiled command: the headline loop (fused tile kernel) + bench.py's own un-fused / fused roofline legs, so that the trace
# holds blend_bwd_kernel<true> launches (the kernel `roofline` is quoted for) next to the fused kernel's
BENCH = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "6", "--no-cpu-baseline", "--refine-iters", "0",
         "--no-extras", "--no-pmc"]


def short(name):
    return name.split("(")[0].replace("void ", "")


def run(args, d):
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3"] + args + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + BENCH
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode != 0:
        print("FAILED", " ".join(cmd), r.stderr[-800:])
    return " ".join(cmd[:len(args) + 1]) + " -- python bench.py " + " ".join(BENCH[2:])


# ---- 1. kernel trace + stats
cmd = run(["--kernel-trace", "--stats"], "/tmp/prof_kt")
for f in glob.glob("/tmp/prof_kt/*kernel_stats.csv") + glob.glob("/tmp/prof_kt/*domain_stats.csv"):
    shutil.copy(f, os.path.join(out, os.path.basename(f).replace("p_", tag + "_bench_")))
rows = list(csv.DictReader(open(glob.glob("/tmp/prof_kt/*kernel_trace.csv")[0])))
rows = sorted((r for r in rows if "sgr::" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
VIEWED = ("tile_scan", "scatter", "blend_fwd", "blend_bwd", "bwd_dense", "zero_heads")   # grid y = views of the batch


def views_of(rows, i):
    """views of launch i: its own grid y when the kernel has a view dimension, else that of the nearest such launch."""
    for d in range(0, 6):
        for j in (i + d, i - d):
            if 0 <= j < len(rows) and any(t in rows[j]["Kernel_Name"] for t in VIEWED):
                return int(rows[j]["Grid_Size_Y"]) // max(1, int(rows[j]["Workgroup_Size_Y"]))
    return 1


per = collections.defaultdict(list)
for i, r in enumerate(rows):
    per[short(r["Kernel_Name"])].append((views_of(rows, i), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
summ = {}
for k, v in per.items():
    g = max(x[0] for x in v)
    sel = [x[1] for x in v if x[0] == g]
    summ[k] = {"views_per_launch": g, "launches": len(sel), "avg_us": round(sum(sel) / len(sel) / 1e3, 2),
               "min_us": round(min(sel) / 1e3, 2), "max_us": round(max(sel) / 1e3, 2)}
step_us = sum(v["avg_us"] for k, v in summ.items() if v["views_per_launch"] > 1)
json.dump({"command": cmd, "note": "per kernel: the batched launches of the timed loop only (views_per_launch = largest batch seen); "
           "durations are back-to-back (start = previous kernel's end), so each includes the ~5 us dispatch gap",
           "sum_of_batched_kernels_us": round(step_us, 1), "kernels": summ},
          open(os.path.join(out, tag + "_kernel_batched_avg.json"), "w"), indent=1)
print(json.dumps({k: v["avg_us"] for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["avg_us"]) if v["views_per_launch"] > 1}), round(step_us, 1))

# ---- 2. PMC passes
NOVIEW = ("sgr::preprocess_fwd_kernel",)
def pmc(counters, d):
    c = run(["--pmc"] + counters + ["--kernel-trace"], d)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(glob.glob(d + "/*counter_collection.csv")[0])):
        if "sgr::" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    res = {}
    for k, cs in agg.items():
        res[k] = {}
        for cn, v in cs.items():
            g = max(x[0] for x in v)
            sel = [x[1] for x in v if x[0] == g]
            if k in NOVIEW:            # grid does not depend on the batch: the 12-view launches are the upper cluster of values
                hi = max(sel)
                sel = [x for x in sel if x > 0.5 * hi]
            res[k][cn] = sum(sel) / len(sel)
            res[k]["launches"] = len(sel)
    return c, res

c1, fetch = pmc(["FETCH_SIZE"], "/tmp/prof_f")
c2, write = pmc(["WRITE_SIZE"], "/tmp/prof_w")
hbm = {}
for k in fetch:
    f, w = fetch[k].get("FETCH_SIZE", 0.0), write.get(k, {}).get("WRITE_SIZE", 0.0)
    hbm[k] = {"FETCH_SIZE_avg_KB_per_launch": round(f, 2), "WRITE_SIZE_avg_KB_per_launch": round(w, 2),
              "launches": fetch[k]["launches"], "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
json.dump({"command": c1 + "   |   " + c2,
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts 128-B requests as 64 B; MI355X_MICROARCH.md HBM)",
           "note": "batched (largest-grid) launches only", "workload": [300000, 640, 480, 12], "kernels": hbm},
          open(os.path.join(out, tag + "_pmc_hbm_bytes.json"), "w"), indent=1)
c3, sq = pmc(["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
              "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"], "/tmp/prof_s")
for k, v in sq.items():
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    v["frac_active"] = round(v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
    v["frac_wait_memory_or_barrier"] = round(v.get("SQ_WAIT_ANY", 0) / wc, 3)
    v["frac_issue_stall"] = round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
    v["valu_insts_per_wave"] = round(v.get("SQ_INSTS_VALU", 0) / max(1.0, v.get("SQ_WAVES", 1)), 1)
    # one SIMD retires one wave64 VALU instruction per quad-cycle: chip-wide VALU-busy time if perfectly spread
    v["valu_quadcycles_per_simd"] = round(v.get("SQ_ACTIVE_INST_VALU", 0) / 1024.0, 1)
json.dump({"command": c3, "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)",
           "note": "batched (largest-grid) launches only; 1024 SIMDs", "kernels": sq},
          open(os.path.join(out, tag + "_pmc_sq.json"), "w"), indent=1)
for name in ("kernel_batched_avg", "pmc_hbm_bytes"):       # what bench.py quotes `roofline.traffic` from
    shutil.copy(os.path.join(out, tag + "_" + name + ".json"), os.path.join(out, "latest_" + name + ".json"))
    shutil.copy(os.path.join(out, tag + "_" + name + ".json"), os.path.join(ROOT, "profiles", "latest_" + name + ".json"))

# ---- 3. the plain bench line (defaults), for tests/test_bench_contract.py and BASELINE.md -- LAST, so that its `roofline.traffic`
# is quoted from the PMC passes above
r0 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], cwd=ROOT, env=env, capture_output=True, text=True)
line = [l for l in r0.stdout.splitlines() if l.startswith("{")]
if line:
    open(os.path.join(out, tag + "_bench_line.json"), "w").write(line[-1] + "\n")
else:
    print("bench.py failed", r0.stderr[-1500:])

print("done", os.listdir(out))
