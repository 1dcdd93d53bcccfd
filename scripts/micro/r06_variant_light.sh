# same-box A/B of library builds on the two bench scenes only (no session): bash scripts/micro/r06_variant_light.sh default a b
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
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
    out[name] = [r["ms_per_step"]] + [round(v, 5) for v in r["kernel_ms"].values()]
print(json.dumps(out))
PY
 done
done
