"""GPU box: HIP-event totals per kernel kind over bench.py's 40-frame session leg (extra.session) -- which kernel a young map's
keyframes spend their time in.   python scripts/micro/session_kernels.py [frames=40]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
sys.argv = [sys.argv[0], "--no-extras", "--no-cpu-baseline", "--refine-iters", "0", "--no-pmc"]
B = bench.Bench(bench.parse())
B.session_leg(frames_n=min(frames, 16), refine_iters=0, warm_frames=0)          # warm-up (allocator, code)
B.lib.sgr_profile_enable(0x7f)
r = B.session_leg(frames_n=frames, refine_iters=0, warm_frames=0)
ms, cnt = (C.c_float * 7)(), (C.c_int64 * 7)()
B.lib.sgr_profile_read(ms, cnt)
B.lib.sgr_profile_enable(0)
names = ["preprocess_fwd", "tile_scan", "scatter", "blend_fused", "blend_fwd", "blend_bwd", "preprocess_bwd_incl_optimiser"]
k = {n: {"total_ms": round(float(ms[i]), 1), "launches": int(cnt[i]), "avg_ms": round(float(ms[i]) / max(1, int(cnt[i])), 4)} for i, n in enumerate(names) if int(cnt[i])}
print(json.dumps({"frames": frames, "keyframes_mapped": r["keyframes_mapped"], "gaussians_final": r["gaussians_final"], "ms_per_keyframe_with_event_overhead": r["ms_per_keyframe"],
                  "ms_init": r["ms_init_keyframe_1050_iterations"], "kernels": k}, indent=1))
