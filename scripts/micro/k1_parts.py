"""GPU box: event time of K1 (preprocess_fwd) on the bench scene under one SGR_DEBUG setting (sgr_common.h: bits 0-7 switch parts of
the kernel off; results are wrong under most of them).  Run once per setting:

    for d in 0 1 3 16 32 64 128; do SGR_DEBUG=$d python scripts/micro/k1_parts.py; done
"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
sa = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
sys.argv = [sys.argv[0], "--no-extras", "--no-cpu-baseline", "--refine-iters", "0"]
B = bench.Bench(bench.parse())
loop, cams = B.build("fused", sa)
B.run_steps(loop, 40)
r = B.profiled(loop, 60, 1 << 0, True)
print(json.dumps({"SGR_DEBUG": int(os.environ.get("SGR_DEBUG", "0")), "scale_add": sa, "preprocess_fwd_ms": round(r[0][0], 5)}))
