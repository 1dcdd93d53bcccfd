mkdir -p gpurun_out/r05
for i in 1 2; do
python scripts/run_session_config1.py --oracle-views 0 --refine 0 --out gpurun_out/r05/sess_default_$i.json > /dev/null 2>&1
SPLAT_SPAN_CACHE=0 SPLAT_VERIFY_ESTIMATES=1 python scripts/run_session_config1.py --oracle-views 0 --refine 0 --out gpurun_out/r05/sess_r4path_$i.json > /dev/null 2>&1
done
python - <<'PY'
import json
for n in ("default_1","r4path_1","default_2","r4path_2"):
    d=json.load(open("gpurun_out/r05/sess_%s.json"%n)); print(n, d["ms_per_mapped_keyframe"], d["gaussians_final"], d["overflow_events"])
PY
