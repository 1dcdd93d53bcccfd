# round 6, call 1: A/B of the tile kernels (round-5 build in splat_slam_amd/lib_r5 vs this tree) + VALU instructions per wave
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
for i in 1 2; do
  SPLAT_HIP_LIB=$GRAFT_REPO_ROOT/splat_slam_amd/lib_r5/libsplat_hip.so python scripts/tile_kernel_times.py light opaque 2>/dev/null | grep scene > gpurun_out/r06a/tile_r5_$i.jsonl
  python scripts/tile_kernel_times.py light opaque 2>/dev/null | grep scene > gpurun_out/r06a/tile_r6_$i.jsonl
done
python scripts/micro/pmc_tile.py gpurun_out/r06a/pmc_tile_opaque_full.json --scale-add 1.6 --passes 0 > gpurun_out/r06a/full.log 2>&1
python scripts/micro/pmc_tile.py gpurun_out/r06a/pmc_tile_light_full.json --passes 0 > gpurun_out/r06a/light.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06a/tile_*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(f.split("/")[-1], d["scene"], d["ms_per_step"], "fwd", d["blend_fwd_ms"], "bwd", d["blend_bwd_ms"], "fused", d["fused_ms"])
PY
tail -2 gpurun_out/r06a/full.log gpurun_out/r06a/light.log
