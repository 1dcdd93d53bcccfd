"""CPU (no GPU needed): would a larger tile per wave pay?  (Round-3 review, item 1a: "evaluate a 16x8 or 16x16 tile per wave".)

For the bench's light scene (scale_add 0) and its opaque one (1.6) and one orbit camera, a sample of 16x16 super tiles is composited
with the oracle's projection, the exact alpha >= 1/255 test and front-to-back termination -- once as four 8x8 tiles (today's waves),
once as two 16x8 tiles (two pixels per lane), once as one 16x16 tile (four pixels per lane) -- and the wave-instruction count of the
fused tile kernel is modelled from the measured per-part costs of the current kernel (profiles/r03_pmc_sq.json: 1 216 VALU
instructions per wave at a mean list of 11; forward walk ~45 per two-splat trip, backward ~95 per (64-lane) iteration of a pixel
pair, ~55 per chunk, ~320 fixed per wave, ~6 per key of the rank sort):

    fixed + sort + forward(list, pixels per lane) + backward(sum over chunks of pixel-pairs / (64 / GW) iterations)

What decides it: a lane is a splat in the backward, and a chunk of width GW costs (pixel pairs of the tile) x GW / 64 iterations. A
larger tile's list is the UNION of its sub-tiles' lists, so every splat is evaluated against pixels of sub-tiles it does not reach.

    python scripts/tile_shape_model.py [gaussians=300000] [super_tiles=60]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import raster_oracle as ro          # noqa: E402  (analysis script: the oracle is the measuring stick here)
from splat_slam_amd import synthetic as syn     # noqa: E402

torch.manual_seed(43)
np.random.seed(43)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
from scipy.spatial import cKDTree               # noqa: E402


def knn_fn(x):
    d, _ = cKDTree(x.numpy()).query(x.numpy(), k=4)
    return torch.as_tensor((d[:, 1:] ** 2).mean(1), dtype=torch.float32)


FIXED, SORT_PER_KEY, FWD_TRIP, BWD_ITER, BWD_CHUNK = 320.0, 6.0, 45.0, 95.0, 55.0


def chunk_plan(n):
    out, end = [], n
    while end > 0:
        gw = 64 if end >= 48 else 32 if end >= 24 else 16 if end >= 12 else 8 if end >= 5 else 4
        out.append(gw)
        end = max(0, end - gw)
    return out


def wave_cost(count, walked, pixels):
    """Modelled VALU instructions of ONE wave that owns `pixels` pixels (64 lanes: pixels / 64 per lane in the forward)."""
    ppl = pixels // 64
    # forward: the T chain is per pixel; with several pixels per lane everything packs across the lane's pixel pairs
    fwd = {1: FWD_TRIP / 2.0, 2: 28.0, 4: 52.0}[ppl] * walked
    sort = SORT_PER_KEY * count if count <= 64 else 12.0 * count          # (register-blocked network beyond a bucket)
    bwd = sum(BWD_ITER * (pixels // 2) * gw / 64.0 + BWD_CHUNK for gw in chunk_plan(walked))
    return FIXED + sort + fwd + bwd


for scale_add in (0.0, 1.6):
    params = syn.room_parameters(N, seed=43, knn_fn=knn_fn, device="cpu")
    intr = syn.INTRINSICS["metric"]
    H, W = intr["H"], intr["W"]
    w2c = torch.as_tensor(syn.orbit_w2c(3, 16), dtype=torch.float32)
    s = ro.make_settings(w2c, intr["fx"], intr["fy"], intr["cx"], intr["cy"], W, H)
    xyz = params["xyz"].float()
    sc = torch.exp(params["scaling"].float() + scale_add)
    rot = torch.nn.functional.normalize(params["rotation"].float())
    op = torch.sigmoid(params["opacity"].float())
    pp = ro.preprocess(xyz, None, op, None, torch.rand(N, 3), sc, rot, None, None, None, s)
    vis = pp.visible.nonzero()[:, 0]
    xy, con, o, dep, rad = pp.xy[vis], pp.conic[vis], pp.opacity[vis], pp.depth[vis], pp.radii[vis].float()
    sgx, sgy = W // 16, H // 16
    rng = np.random.default_rng(0)
    tot = {"8x8": 0.0, "16x8": 0.0, "16x16": 0.0}
    lists = {"8x8": [], "16x8": [], "16x16": []}
    evals = {"8x8": 0, "16x8": 0, "16x16": 0}
    live = 0
    for st in rng.choice(sgx * sgy, NS, replace=False):
        sx, sy = st % sgx, st // sgx
        x0, y0 = 16 * sx, 16 * sy
        # candidates: the reference's 16x16 rectangle test
        m = (xy[:, 0] + rad >= x0 - 1) & (xy[:, 0] - rad < x0 + 17) & (xy[:, 1] + rad >= y0 - 1) & (xy[:, 1] - rad < y0 + 17)
        idx = m.nonzero()[:, 0]
        if idx.numel() == 0:
            for k in tot:
                tot[k] += {"8x8": 4, "16x8": 2, "16x16": 1}[k] * FIXED
            continue
        idx = idx[torch.argsort(dep[idx])]
        px = (x0 + torch.arange(16)).float()[None, :].expand(16, 16).reshape(-1)
        py = (y0 + torch.arange(16)).float()[:, None].expand(16, 16).reshape(-1)
        dx, dy = xy[idx, 0:1] - px[None], xy[idx, 1:2] - py[None]
        power = -0.5 * (con[idx, 0:1] * dx * dx + con[idx, 2:3] * dy * dy) - con[idx, 1:2] * dx * dy
        alpha = torch.clamp(o[idx, None] * torch.exp(power), max=0.99)
        ok = (power <= 0) & (alpha >= 1 / 255.0)
        T = torch.ones(256)
        done = torch.zeros(256, dtype=torch.bool)
        contrib = torch.zeros_like(ok)
        for j in range(idx.numel()):
            a = torch.where(ok[j], alpha[j], torch.zeros(256))
            test = T * (1 - a)
            term = ok[j] & ~done & (test < 1e-4)
            c = ok[j] & ~done & ~term
            done |= term
            contrib[j] = c
            T = torch.where(c, test, T)
        live += int(contrib.sum())
        okg, cg = ok.reshape(-1, 16, 16), contrib.reshape(-1, 16, 16)
        for name, (th, tw) in (("8x8", (8, 8)), ("16x8", (8, 16)), ("16x16", (16, 16))):
            for ty in range(0, 16, th):
                for tx in range(0, 16, tw):
                    binned = okg[:, ty:ty + th, tx:tx + tw].reshape(okg.shape[0], -1).any(1)          # the exact footprint test keeps these
                    used = cg[:, ty:ty + th, tx:tx + tw].reshape(cg.shape[0], -1).any(1) & binned
                    count = int(binned.sum())
                    sel = binned.nonzero()[:, 0]
                    walked = int((used[sel].nonzero().max() + 1)) if bool(used[sel].any()) else 0
                    lists[name].append(walked)
                    evals[name] += walked * th * tw
                    tot[name] += wave_cost(count, walked, th * tw)
    base = tot["8x8"]
    print(f"scene scale_add {scale_add}: {vis.numel()} visible Gaussians, {NS} super tiles sampled")
    for name in tot:
        L = np.array(lists[name])
        print(f"  {name:6s} waves per super tile {len(L) / NS:.0f}  mean walked list {L.mean():6.1f}  (pixel, splat) evaluations x{evals[name] / evals['8x8']:.2f} "
              f" live fraction {live / max(1, evals[name]):.2f}  modelled VALU instructions x{tot[name] / base:.3f}")
