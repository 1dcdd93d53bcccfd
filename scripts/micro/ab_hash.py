"""GPU box: a fingerprint of what N fused mapping iterations of the bench scene leave behind (sha256 over every parameter tensor
and the per-iteration losses) plus the time of the tile kernel -- the loop is bitwise run-to-run deterministic, so two BUILDS of
libsplat_hip.so that claim the same arithmetic must print the same hash:

    SPLAT_HIP_LIB=/path/to/other/libsplat_hip.so python scripts/micro/ab_hash.py [--scale-add 1.6] [--steps 40]
"""
import argparse
import hashlib
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale-add", type=float, default=0.0)
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
sys.argv = [sys.argv[0], "--no-extras", "--no-cpu-baseline", "--refine-iters", "0"]
B = bench.Bench(bench.parse())
loop, cams = B.build("fused", a.scale_add)
B.run_steps(loop, a.steps)
torch.cuda.synchronize()
h = hashlib.sha256()
gm = loop.gaussians
for name in ("_xyz", "_features_dc", "_scaling", "_rotation", "_opacity"):
    h.update(getattr(gm, name).detach().cpu().numpy().tobytes())
r = B.profiled(loop, 40, (1 << bench.PK_FUSED) | 1 | (1 << 6), True)
el, _ = B.timed(loop, 100)
print(json.dumps({"lib": os.environ.get("SPLAT_HIP_LIB", "in-tree"), "scale_add": a.scale_add, "sha256": h.hexdigest()[:16],
                  "fused_ms": round(r[bench.PK_FUSED][0], 5), "k1_ms": round(r[0][0], 5), "dense_adam_ms": round(r[6][0], 5), "ms_per_step": round(1e3 * el / 100, 4)}))
