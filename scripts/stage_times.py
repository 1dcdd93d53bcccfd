"""Per-stage HIP-event times of the bench workload (timing experiments; SGR_DEBUG switches stages off)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for dbg in sys.argv[1:] or ["0"]:
    env = dict(os.environ, SGR_DEBUG=dbg)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "5", "--no-cpu-baseline",
                          "--refine-iters", "0", "--profile-all"], env=env, capture_output=True, text=True).stdout
    try:
        d = json.loads(out.strip().splitlines()[-1])
        print("dbg", dbg, "ms/step", d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["kernel_ms"].items() if v})
    except Exception as e:
        print("dbg", dbg, "failed", e, out[-300:])
