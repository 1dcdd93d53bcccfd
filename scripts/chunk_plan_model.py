"""CPU (no GPU needed): what would the backward's loop cost under other chunk plans?  For the bench scene at --scale-add (1.6 = the
opaque, converged-style map) and one orbit camera it walks a sample of 8x8 tiles with the oracle's projection, the exact
alpha >= 1/255 test and front-to-back termination, takes every pixel pair's LAST contributor and models the VALU instructions of
tile_backward() (sgr_blend.hip) for

    current   the binary chunk plan cut from the far end; a 64-wide chunk skips a pixel pair that ended before the chunk
    alive-W   chunks of at most W splats (W = 64 / 32 / 16), every chunk only iterates over the pixel pairs that are still alive
              at its first splat (pairs whose last contributor lies in front of the chunk contribute exact zeros)

    python scripts/chunk_plan_model.py [scale_add=1.6] [gaussians=300000] [tiles=400]

Instruction costs per iteration / chunk are the measured ones of DESIGN.md 3 (83 / 80 / 75 / 68 / 62 at 64 / 32 / 16 / 8 / 4 lanes,
~95 per chunk for prologue + epilogue)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from splat_slam_amd import synthetic as syn
from oracle import raster_oracle as ro
torch.manual_seed(43); np.random.seed(43)
scale_add = float(sys.argv[1]) if len(sys.argv) > 1 else 1.6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
NT = int(sys.argv[3]) if len(sys.argv) > 3 else 400
from scipy.spatial import cKDTree
def knn_fn(x):
    d, _ = cKDTree(x.numpy()).query(x.numpy(), k=4)
    return torch.as_tensor((d[:, 1:] ** 2).mean(1), dtype=torch.float32)
params = syn.room_parameters(N, seed=43, knn_fn=knn_fn, device="cpu")
intr = syn.INTRINSICS["metric"]
H, W = intr["H"], intr["W"]
w2c = torch.as_tensor(syn.orbit_w2c(3, 16), dtype=torch.float32)
s = ro.make_settings(w2c, intr["fx"], intr["fy"], intr["cx"], intr["cy"], W, H)
xyz = params["xyz"].float(); sc = torch.exp(params["scaling"].float() + scale_add)
rot = torch.nn.functional.normalize(params["rotation"].float()); op = torch.sigmoid(params["opacity"].float())
col = torch.rand(N, 3)
pp = ro.preprocess(xyz, None, op, None, col, sc, rot, None, None, None, s)
vis = pp.visible.nonzero()[:, 0]
rad = pp.radii[vis].float(); xy = pp.xy[vis]; con = pp.conic[vis]; o = pp.opacity[vis]; dep = pp.depth[vis]
rect = torch.stack([torch.clamp(torch.trunc((xy[:, 0] - rad) / 8), 0, W // 8), torch.clamp(torch.trunc((xy[:, 1] - rad) / 8), 0, H // 8),
                    torch.clamp(torch.trunc((xy[:, 0] + rad + 7) / 8), 0, W // 8), torch.clamp(torch.trunc((xy[:, 1] + rad + 7) / 8), 0, H // 8)], 1).long()
gx, gy = W // 8, H // 8
rng = np.random.default_rng(0)
tiles = rng.choice(gx * gy, min(NT, gx * gy), replace=False)
COST = {64: 83, 32: 80, 16: 75, 8: 68, 4: 62}
CHUNK = 95


def width(end):
    return 64 if end >= 61 else (32 if end >= 29 else (16 if end >= 13 else (8 if end >= 5 else 4)))


def current(eff, keys):
    tot, end = 0.0, eff
    while end > 0:
        gw = width(end); start = max(0, end - gw)
        it = 32 * gw // 64
        if gw == 64:
            dead = int((keys <= start).sum())
            tot += (it - dead) * COST[gw] + dead * 12
        else:
            tot += it * COST[gw]
        tot += CHUNK
        end = start
    return tot


def alive(eff, keys, wmax):
    tot, end = 0.0, eff
    while end > 0:
        gw = min(width(end), wmax); start = max(0, end - gw)
        pp_ = 64 // gw
        n_alive = int((keys > start).sum())
        it = -(-n_alive // pp_)
        tot += it * (COST[gw] + 3) + CHUNK + 6
        end = start
    return tot


res = {"current": 0.0, "alive-64": 0.0, "alive-32": 0.0, "alive-16": 0.0}
lens = []
for t in tiles:
    tx, ty = t % gx, t // gx
    m = (rect[:, 0] <= tx) & (rect[:, 2] > tx) & (rect[:, 1] <= ty) & (rect[:, 3] > ty)
    idx = m.nonzero()[:, 0]
    if idx.numel() == 0:
        continue
    idx = idx[torch.argsort(dep[idx])]
    px = (tx * 8 + torch.arange(8)).float()[None, :].expand(8, 8).reshape(-1)
    py = (ty * 8 + torch.arange(8)).float()[:, None].expand(8, 8).reshape(-1)
    dx = xy[idx, 0:1] - px[None]; dy = xy[idx, 1:2] - py[None]
    power = -0.5 * (con[idx, 0:1] * dx * dx + con[idx, 2:3] * dy * dy) - con[idx, 1:2] * dx * dy
    alpha = torch.clamp(o[idx, None] * torch.exp(power), max=0.99)
    ok = (power <= 0) & (alpha >= 1 / 255.)
    keep = ok.any(1)
    ok = ok[keep]; alpha = alpha[keep]
    n = ok.shape[0]
    if n == 0:
        continue
    T = torch.ones(64); done = torch.zeros(64, dtype=torch.bool); last = torch.zeros(64, dtype=torch.long)
    for j in range(n):
        a = torch.where(ok[j], alpha[j], torch.zeros(64))
        test = T * (1 - a)
        term = ok[j] & ~done & (test < 1e-4)
        c = ok[j] & ~done & ~term
        done |= term
        last = torch.where(c, torch.full_like(last, j + 1), last)
        T = torch.where(c, test, T)
    eff = int(last.max())
    if eff == 0:
        continue
    keys = last.reshape(32, 2).max(1).values.numpy()
    lens.append(eff)
    res["current"] += current(eff, keys)
    for w in (64, 32, 16):
        res["alive-%d" % w] += alive(eff, keys, w)
L = np.array(lens)
print("scale_add", scale_add, "tiles", len(L), "mean walked list", L.mean(), "max", L.max())
for k, v in res.items():
    print("%-9s %8.1f VALU instructions per tile (backward loops)   x%.3f" % (k, v / len(L), v / res["current"]))
