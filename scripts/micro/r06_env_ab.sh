# same-box A/B of ENVIRONMENT switches of the library on the two bench scenes + the 40-frame session:
#   bash scripts/micro/r06_env_ab.sh "" "SGR_DENSE_QUAD=1" ...
cd $GRAFT_REPO_ROOT
for i in 1 2; do
 for envs in "$@"; do
  env $envs python - <<PY
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
sys.argv = [sys.argv[0], "--no-extras", "--no-cpu-baseline", "--refine-iters", "0", "--no-pmc"]
B = bench.Bench(bench.parse())
out = {"env": "$envs"}
for name, sa in (("light", 0.0), ("opaque", 1.6)):
    r = B.scene_leg(sa, steps=60)
    out[name] = [r["ms_per_step"]] + [round(v, 5) for v in r["kernel_ms"].values()]
s = B.session_leg(frames_n=40, refine_iters=0)
out["session"] = [s["ms_per_keyframe"], s["psnr_all_keyframes_mean"]]
print(json.dumps(out))
PY
 done
done
