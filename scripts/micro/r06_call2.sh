# round 6, call 2: full -m gpu suite on the termination-tracking build + A/B against the round-5 build + VALU instructions per wave
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for i in 1 2; do
  SPLAT_HIP_LIB=$GRAFT_REPO_ROOT/splat_slam_amd/lib_r5/libsplat_hip.so python scripts/tile_kernel_times.py light opaque 2>/dev/null | grep scene > gpurun_out/r06b/tile_r5_$i.jsonl
  python scripts/tile_kernel_times.py light opaque 2>/dev/null | grep scene > gpurun_out/r06b/tile_r6_$i.jsonl
done
python scripts/micro/pmc_tile.py gpurun_out/r06b/pmc_tile_opaque_full.json --scale-add 1.6 --passes 0 > gpurun_out/r06b/full.log 2>&1
python scripts/micro/pmc_tile.py gpurun_out/r06b/pmc_tile_light_full.json --passes 0 > gpurun_out/r06b/light.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06b/tile_*.jsonl")):
    for l in open(f):
        d=json.loads(l); print(f.split("/")[-1], d["scene"], d["ms_per_step"], "fwd", d["blend_fwd_ms"], "bwd", d["blend_bwd_ms"], "fused", d["fused_ms"])
for f in ("pmc_tile_opaque_full.json","pmc_tile_light_full.json"):
    d=json.load(open("gpurun_out/r06b/"+f))
    for k,v in d["kernels"].items():
        print(f, k, v["per_wave"].get("SQ_INSTS_VALU"))
PY
