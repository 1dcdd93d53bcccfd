"""`simple_knn._C.distCUDA2` backed by the gfx950 kernel `sknn_dist2` (include/splat_hip.h).

Reference call site: /root/reference/thirdparty/gaussian_splatting/scene/gaussian_model.py:194-200.
Upstream (camenduru/simple-knn) finds the exact 3 nearest neighbours with a Morton-ordered box search.  Here: up to 16 k
points (what the mapper feeds it: one keyframe's seeds) the exact search is brute force through LDS tiles, beyond that a
uniform-grid search that widens its box until the third-nearest distance is proven -- both exact, order-independent and
deterministic (csrc/sgr_aux.hip).
"""
import ctypes as C

import torch

from splat_slam_amd import _native as nat


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if not points.is_cuda:
        raise RuntimeError("simple_knn (MI355X build): points must be a GPU tensor; there is no CPU path")
    pts = points.detach().float().contiguous()
    n = pts.shape[0]
    out = torch.empty((n,), dtype=torch.float32, device=pts.device)
    if n == 0:
        return out
    lib = nat.lib()
    nbytes = lib.sknn_scratch_bytes(n)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        nat.check(lib.sknn_dist2(pts.data_ptr(), n, out.data_ptr(), scratch.data_ptr(), nbytes,
                                 torch.cuda.current_stream(pts.device).cuda_stream), "sknn_dist2")
    return out
