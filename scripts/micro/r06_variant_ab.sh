# round 6: same-box A/B of library BUILDS (splat_slam_amd/lib_<name>/libsplat_hip.so; "default" = the tree's own build), alternating twice.
#   bash scripts/micro/r06_variant_ab.sh default p1 p4
cd $GRAFT_REPO_ROOT
for i in 1 2; do
 for lib in "$@"; do
  if [ $lib = default ]; then unset SPLAT_HIP_LIB; else export SPLAT_HIP_LIB=$GRAFT_REPO_ROOT/splat_slam_amd/lib_$lib/libsplat_hip.so; fi
  python - <<PY
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
sys.argv = [sys.argv[0], "--no-extras", "--no-cpu-baseline", "--refine-iters", "0", "--no-pmc"]
B = bench.Bench(bench.parse())
out = {"lib": "$lib"}
for name, sa in (("light", 0.0), ("opaque", 1.6)):
    r = B.scene_leg(sa, steps=60)
    out[name] = {"ms_per_step": r["ms_per_step"], "kernel_ms": r["kernel_ms"]}
s = B.session_leg(frames_n=40, refine_iters=0)
out["session_ms_per_keyframe"] = s["ms_per_keyframe"]; out["gaussians"] = s["gaussians_final"]; out["psnr"] = s["psnr_all_keyframes_mean"]
print(json.dumps(out))
PY
 done
done
