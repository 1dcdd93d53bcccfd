"""GPU box: the bench session (bench.py extra.session: 40 tracker frames, default hyper-parameters) under the host-path switches of
FusedMappingLoop, alternating in ONE call so that box speed cancels.
    python scripts/session_ab.py [--repeat 2] [--out x.json]
Variants (environment of a child process each): round-4 host path (SPLAT_SPAN_CACHE=0 SPLAT_VERIFY_ESTIMATES=1) vs round 5 (launch structs dropped at every keyframe: SPLAT_KEYFRAME_STRUCTS=0) vs round 6.
--full: the 160-frame session (bench.py extra.session_full) instead."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--repeat", type=int, default=2)
ap.add_argument("--out", default=None)
ap.add_argument("--full", action="store_true")
ap.add_argument("--only", default=None, help="comma-separated variant names")
a = ap.parse_args()
CHILD = ("import json, sys; sys.path.insert(0, %r); import bench; sys.argv=['bench.py']; B = bench.Bench(bench.parse()); "
         "r = B.session_leg(refine_iters=0%s); print('RESULT ' + json.dumps({k: r[k] for k in ('ms_per_keyframe', 'ms_per_keyframe_second_half', "
         "'gaussians_final', 'psnr_all_keyframes_mean', 'overflow_events')}))" % (ROOT, ", frames_n=160, step_of=160, warm_frames=0" if a.full else ""))
VARIANTS = {"round4_host_path": {"SPLAT_SPAN_CACHE": "0", "SPLAT_VERIFY_ESTIMATES": "1"},
            "span_cache_only": {"SPLAT_SPAN_CACHE": "1", "SPLAT_VERIFY_ESTIMATES": "1"},
            "no_verify_only": {"SPLAT_SPAN_CACHE": "0", "SPLAT_VERIFY_ESTIMATES": "0"},
            "round5_default": {"SPLAT_KEYFRAME_STRUCTS": "0", "SPLAT_SNAPSHOT_STORE": "0"},
            "no_snapshot_store": {"SPLAT_SNAPSHOT_STORE": "0"},
            "round6_default": {}}
if a.only:
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in a.only.split(",")}
res = {k: [] for k in VARIANTS}
for rep in range(a.repeat):
    for name, env in VARIANTS.items():
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(name, "FAILED", r.stderr[-800:])
            continue
        res[name].append(json.loads(line[-1][7:]))
        print(name, res[name][-1], flush=True)
out = {"what": "bench.py extra.session (40 tracker frames, 12-frame throw-away session first) per variant, alternating in one call",
       "variants": {k: {"env": VARIANTS[k], "runs": v, "ms_per_keyframe_mean": round(sum(x["ms_per_keyframe"] for x in v) / max(1, len(v)), 3)} for k, v in res.items()}}
print(json.dumps({k: v["ms_per_keyframe_mean"] for k, v in out["variants"].items()}))
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
