import os, sys, collections
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from splat_slam_amd import synthetic as syn
from splat_slam_amd.fused import FusedMappingLoop
from splat_slam_amd.session import MappingSession
dev = torch.device("cuda:0"); intr = syn.INTRINSICS["metric"]
torch.manual_seed(43); np.random.seed(43)
frames = syn.keyframe_stream(160, intr, dev, n_world=400000, seed=43, sweep_deg=360.0 * 159 / 160)[:60]
loop = FusedMappingLoop(syn.DEFAULT_CONFIG, device=dev); sess = MappingSession(loop, intr)
why = collections.Counter()
orig = FusedMappingLoop._map_view
def spy(self, cam, initialization=False, images=True, slot=False):
    vb = self._view(cam)
    key = (cam._version, self._cap, id(cam.exposure_a), id(cam.original_image), vb.gt_depth.data_ptr(), initialization, self.keyframe_optimizers is not None)
    hit = (vb.mv or {}).get((images, slot))
    if vb.mv is None: why["mv None"] += 1
    elif hit is None: why["no entry"] += 1
    elif hit[0] != key:
        names = ["version", "cap", "exp_a id", "image id", "gt_depth ptr", "init", "kfopt"]
        why["key: " + ",".join(n for n, a, b in zip(names, hit[0], key) if a != b)] += 1
    else: why["hit"] += 1
    return orig(self, cam, initialization, images, slot)
FusedMappingLoop._map_view = spy
for i, f in enumerate(frames):
    if i == 40: why.clear()
    sess.process(*f)
print(dict(why), "dirty gen", loop._gen)
