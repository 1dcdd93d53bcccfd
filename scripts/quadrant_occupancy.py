"""CPU (no GPU needed): would sub-tile footprint masks pay?  For the bench's opaque scene (or any --scale-add) and one orbit camera
it walks a sample of 8x8 tiles with the oracle's projection, the exact alpha >= 1/255 test and front-to-back termination, and
reports which part of every walked (tile, splat) pair's 4x4 quadrants / pixel rows / pixel pairs / pixels receives a
contribution, plus a model of the backward's loop iterations under the current chunk plan vs. per-quadrant lists.

    python scripts/quadrant_occupancy.py [scale_add=1.6] [gaussians=300000]

Result on the opaque scene (300 k, +1.6): quadrants 0.81, pixels 0.68 of the walked pairs are live; per-quadrant lists would run
0.985x the iterations of the current plan -- the splats of a converged map are larger than an 8x8 tile, there is nothing to skip."""
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
from scipy.spatial import cKDTree
def knn_fn(x):
    d,_=cKDTree(x.numpy()).query(x.numpy(),k=4)
    return torch.as_tensor((d[:,1:]**2).mean(1),dtype=torch.float32)
params = syn.room_parameters(N, seed=43, knn_fn=knn_fn, device="cpu")
intr = syn.INTRINSICS["metric"]
H, W = intr["H"], intr["W"]
w2c = syn.orbit_w2c(3, 16)
w2c = torch.as_tensor(w2c, dtype=torch.float32)
s = ro.make_settings(w2c, intr["fx"], intr["fy"], intr["cx"], intr["cy"], W, H)
xyz = params["xyz"].float(); sc = torch.exp(params["scaling"].float() + scale_add)
rot = torch.nn.functional.normalize(params["rotation"].float()); op = torch.sigmoid(params["opacity"].float())
col = torch.rand(N, 3)
pp = ro.preprocess(xyz, None, op, None, col, sc, rot, None, None, None, s)
vis = pp.visible.nonzero()[:, 0]
print("visible", vis.numel())
rad=pp.radii[vis].float(); _xy=pp.xy[vis]; rect=torch.stack([torch.clamp(torch.trunc((_xy[:,0]-rad)/8),0,W//8),torch.clamp(torch.trunc((_xy[:,1]-rad)/8),0,H//8),torch.clamp(torch.trunc((_xy[:,0]+rad+7)/8),0,W//8),torch.clamp(torch.trunc((_xy[:,1]+rad+7)/8),0,H//8)],1).long(); xy = pp.xy[vis]; con = pp.conic[vis]; o = pp.opacity[vis]; dep = pp.depth[vis] if hasattr(pp, 'depth') else pp[3][vis]
gx, gy = W // 8, H // 8
rng = np.random.default_rng(0)
tiles = rng.choice(gx * gy, 400, replace=False)
tot_tile = tot_q = tot_row = tot_pp = tot_px = 0; tot_rect = 0
lens = []; qlens = []
for t in tiles:
    tx, ty = t % gx, t // gx
    m = (rect[:, 0] <= tx) & (rect[:, 2] > tx) & (rect[:, 1] <= ty) & (rect[:, 3] > ty)
    idx = m.nonzero()[:, 0]
    if idx.numel() == 0: continue
    order = torch.argsort(dep[idx]); idx = idx[order]
    px = (tx * 8 + torch.arange(8)).float()[None, :].expand(8, 8).reshape(-1)
    py = (ty * 8 + torch.arange(8)).float()[:, None].expand(8, 8).reshape(-1)
    dx = xy[idx, 0:1] - px[None]; dy = xy[idx, 1:2] - py[None]
    power = -0.5 * (con[idx, 0:1] * dx * dx + con[idx, 2:3] * dy * dy) - con[idx, 1:2] * dx * dy
    alpha = torch.clamp(o[idx, None] * torch.exp(power), max=0.99)
    ok = (power <= 0) & (alpha >= 1 / 255.)
    tot_rect += idx.numel()
    # footprint-level list (what the bin test keeps): any pixel ok
    keep = ok.any(1)
    ok = ok[keep]; alpha = alpha[keep]
    n = ok.shape[0]
    if n == 0: continue
    # forward walk with termination
    T = torch.ones(64); contrib = torch.zeros_like(ok)
    done = torch.zeros(64, dtype=torch.bool)
    for j in range(n):
        a = torch.where(ok[j], alpha[j], torch.zeros(64))
        test = T * (1 - a)
        term = ok[j] & ~done & (test < 1e-4)
        c = ok[j] & ~done & ~term
        done |= term
        contrib[j] = c
        T = torch.where(c, test, T)
    used = contrib.any(1)
    last = used.nonzero().max().item() + 1 if used.any() else 0
    cw = contrib[:last]
    lens.append(last)
    tot_tile += last
    q = cw.reshape(last, 2, 4, 2, 4).permute(0, 1, 3, 2, 4).reshape(last, 4, 16).any(2)
    tot_q += q.sum().item()
    qlens.append(q.sum(0).tolist())
    tot_row += cw.reshape(last, 8, 8).any(2).sum().item()
    tot_pp += cw.reshape(last, 32, 2).any(2).sum().item()
    tot_px += cw.sum().item()
print("tiles", len(lens), "mean rect-list", tot_rect / len(lens), "mean walked list", tot_tile / len(lens))
print("quadrant fraction", tot_q / (4 * tot_tile), " row fraction", tot_row / (8 * tot_tile), " pixel-pair fraction", tot_pp / (32 * tot_tile), " pixel fraction", tot_px / (64 * tot_tile))
ql = np.array(qlens); L = np.array(lens)
print("mean max-quadrant-list / list", (ql.max(1) / np.maximum(L, 1)).mean(), " mean quadrant list", ql.mean(), "max", ql.max(), "list max", L.max())
# iteration model backward: current chunk plan vs quadrant plan
def plan_iters(n, per_chunk_full):
    it = 0; end = n
    while end > 0:
        gw = 64 if end >= 48 else 32 if end >= 24 else 16 if end >= 12 else 8 if end >= 5 else 4
        it += per_chunk_full * gw // 64 + 1.3   # + epilogue ~ 1.3 iterations
        end = max(0, end - gw)
    return it
cur = sum(plan_iters(n, 32) for n in L)
quad = sum(sum(plan_iters(int(x), 8) for x in row) for row in ql)
print("backward iteration model: current", cur / len(L), "quadrant", quad / len(L), "ratio", quad / cur)
print("forward model: current", L.mean(), "row-parallel quadrant walk", ql.max(1).mean())
