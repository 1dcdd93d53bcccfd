"""GPU box: SQ counter passes over bench.py's timed loop, reduced to the fused tile kernel (and K1): what the wave's time is made of
beyond the VALU instruction count -- LDS instructions / busy cycles / bank conflicts, scalar and branch instructions, transcendental
share.  Counters in their own runs (--pmc + --kernel-trace only), from /tmp (MI355X_MICROARCH.md).

    python scripts/micro/pmc_tile.py <out.json> [--scale-add 1.6] [--passes 0]
    SGR_DEBUG=2048 | 4096 ...  (bit 11: no walk, no backward; bit 12: no backward): the kernel by regions
"""
import collections, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1]
extra = sys.argv[2:]
only = None                      # --passes 0,2: a subset of the counter passes (the SGR_DEBUG region runs only need the first)
if "--passes" in extra:
    i = extra.index("--passes")
    only = [int(x) for x in extra[i + 1].split(",")]
    extra = extra[:i] + extra[i + 2:]
env = dict(os.environ, TMPDIR="/tmp")
BENCH = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "6", "--no-cpu-baseline", "--refine-iters", "0", "--no-extras", "--no-pmc"] + extra
PASSES = [["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INST_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_SALU", "SQ_INSTS_LDS"],
          ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INSTS_BRANCH"],
          ["SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU_CVT"]]
res = collections.defaultdict(dict)
for n, counters in enumerate(PASSES):
    if only is not None and n not in only:
        continue
    d = "/tmp/pmc_tile_%d" % n
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + BENCH
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode != 0:
        print("FAILED", r.stderr[-600:])
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(glob.glob(d + "/*counter_collection.csv")[0])):
        k = row["Kernel_Name"]
        if "blend_fwd_kernel<512, true>" in k or "preprocess_fwd_kernel" in k or "blend_bwd_kernel<true>" in k or "blend_fwd_kernel<512, false>" in k:
            agg[k.split("(")[0].replace("void ", "")][row["Counter_Name"]].append((int(row["Grid_Size"]), float(row["Counter_Value"])))
    for k, cs in agg.items():
        for cn, v in cs.items():
            g = max(x[0] for x in v)
            sel = [x[1] for x in v if x[0] == g]
            hi = max(sel)
            sel = [x for x in sel if x > 0.5 * hi] or sel
            res[k][cn] = round(sum(sel) / len(sel), 1)
for k, v in res.items():
    w = v.get("SQ_WAVES", 1.0) or 1.0
    v["per_wave"] = {c: round(x / w, 1) for c, x in v.items() if isinstance(x, float) and c != "SQ_WAVES"}
json.dump({"SGR_DEBUG": int(os.environ.get("SGR_DEBUG", "0")), "command": "rocprofv3 --pmc <pass> --kernel-trace -- python bench.py " + " ".join(BENCH[2:]), "passes": PASSES,
           "note": "batched (largest-grid) launches only, averages per launch; per_wave = / SQ_WAVES", "kernels": res}, open(out, "w"), indent=1)
for k, v in res.items():
    print(k, json.dumps(v["per_wave"]))
