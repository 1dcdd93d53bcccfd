"""Helpers for the -m gpu parity tests: run the HIP path through the drop-in package and the oracle side by side."""
import torch

from oracle import raster_oracle as O

GRAD_KEYS = ["means3D", "means2D", "opacities", "shs", "scales", "rotations", "theta", "rho"]


def hip_settings(s, dev):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
    return GaussianRasterizationSettings(
        image_height=s.image_height, image_width=s.image_width, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=f(s.bg),
        scale_modifier=s.scale_modifier, viewmatrix=f(s.viewmatrix), projmatrix=f(s.projmatrix),
        projmatrix_raw=f(s.projmatrix_raw), sh_degree=s.sh_degree, campos=f(s.campos), prefiltered=False, debug=False)


def hip_depth_keys(saved, capacity, n, H, W, radii):
    """fp32 view-space depths the forward that filled `saved` sorted by (sgr_query_depth_keys), on the CPU."""
    import ctypes as C
    from splat_slam_amd import _native as nat
    out = torch.empty(n, dtype=torch.float32, device=radii.device)
    ws = nat.SgrWorkspace(saved.data_ptr(), saved.numel(), None, 0, int(capacity))
    nat.check(nat.lib().sgr_query_depth_keys(C.byref(ws), n, H, W, radii.data_ptr(), out.data_ptr(),
                                             torch.cuda.current_stream(radii.device).cuda_stream), "sgr_query_depth_keys")
    torch.cuda.synchronize()
    return out.cpu()


def check_depth_keys(keys, radii, ref_depth, ulps=8):
    """The override may only re-order near ties: every key must equal the oracle's depth to a few fp32 ulp."""
    vis = radii > 0
    k, d = keys[vis].double(), ref_depth.detach()[vis].double()
    rel = ((k - d).abs() / d.abs()).max().item() if k.numel() else 0.0
    assert rel <= ulps * 2.0 ** -24, f"HIP depth keys differ from the oracle's depths by {rel / 2.0 ** -24:.1f} ulp"


def run_hip(inp, s, wc=None, wd=None, dev="cuda:0", want_depth_keys=False):
    """inp: dict of CPU tensors (any float dtype). Returns (outputs on CPU, grads dict on CPU or None[, depth keys])."""
    from diff_gaussian_rasterization import GaussianRasterizer
    x = {k: v.detach().to(device=dev, dtype=torch.float32).requires_grad_(wc is not None) for k, v in inp.items()}
    rast = GaussianRasterizer(raster_settings=hip_settings(s, dev))
    out = rast(means3D=x["means3D"], means2D=x["means2D"], shs=x.get("shs"), colors_precomp=x.get("colors_precomp"),
               opacities=x["opacities"], scales=x.get("scales"), rotations=x.get("rotations"),
               cov3D_precomp=x.get("cov3D_precomp"), theta=x.get("theta"), rho=x.get("rho"))
    grads = None
    keys = None
    if want_depth_keys:
        from diff_gaussian_rasterization import saved_block_of
        saved, cap = saved_block_of(out[0])
        keys = hip_depth_keys(saved, cap, x["means3D"].shape[0], s.image_height, s.image_width, out[1])
    if wc is not None:
        loss = (out[0] * wc.to(dev).float()).sum() + (out[2] * wd.to(dev).float()).sum()
        loss.backward()
        grads = {k: (v.grad.detach().cpu() if v.grad is not None else None) for k, v in x.items()}
    torch.cuda.synchronize()
    res = [o.detach().cpu() for o in out]
    return (res, grads, keys) if want_depth_keys else (res, grads)


def run_oracle(inp, s, wc=None, wd=None, dtype=torch.float64, depth_sort_key=None, knife=None):
    x = {k: v.detach().to(dtype).requires_grad_(wc is not None) for k, v in inp.items()}
    s2 = s._replace(bg=s.bg.to(dtype), viewmatrix=s.viewmatrix.to(dtype), projmatrix=s.projmatrix.to(dtype),
                    projmatrix_raw=s.projmatrix_raw.to(dtype), campos=s.campos.to(dtype))
    out = O.rasterize(x["means3D"], x["means2D"], x["opacities"], shs=x.get("shs"),
                      colors_precomp=x.get("colors_precomp"), scales=x.get("scales"), rotations=x.get("rotations"),
                      cov3D_precomp=x.get("cov3D_precomp"), theta=x.get("theta"), rho=x.get("rho"), settings=s2,
                      depth_sort_key=depth_sort_key, knife=knife)
    grads = None
    if wc is not None:
        loss = (out[0] * wc.to(dtype)).sum() + (out[2] * wd.to(dtype)).sum()
        loss.backward()
        grads = {k: (v.grad.detach() if v.grad is not None else None) for k, v in x.items()}
    return [o.detach() for o in out], grads


def to_fp32_inputs(inp, s):
    """Round every input to fp32 once so that both sides see bit-identical numbers."""
    inp32 = {k: v.float().double() for k, v in inp.items()}
    s32 = s._replace(bg=s.bg.float(), viewmatrix=s.viewmatrix.float(), projmatrix=s.projmatrix.float(),
                     projmatrix_raw=s.projmatrix_raw.float(), campos=s.campos.float(),
                     tanfovx=float(torch.tensor(s.tanfovx, dtype=torch.float32)),
                     tanfovy=float(torch.tensor(s.tanfovy, dtype=torch.float32)))
    return inp32, s32


def rel_linf(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def outlier_report(a, b, rel):
    """(#elements off by more than rel*max|b|, max abs err, max|b|)."""
    a, b = a.double(), b.double()
    m = b.abs().max().clamp_min(1e-30)
    e = (a - b).abs()
    return int((e > rel * m).sum()), e.max().item(), m.item()


def knife_ids(knife, n):
    """Boolean [n]: Gaussians the oracle saw within KNIFE_BAND of a cut-off at some pixel (raster_oracle.knife_edge_gaussians)."""
    m = torch.zeros(n, dtype=torch.bool)
    for ids in knife.get("gaussians", []):
        m[ids] = True
    return m


def move_off_knife_edges(inp, s, max_rounds=12):
    """Nudges the Gaussians that sit on a cut-off until no decision of the view is within the oracle's knife bands of its
    threshold: activated opacities by 0.2-2 % (alpha >= 1/255, T thresholds), and -- for the integer decisions of the
    projection (16-pixel tile rectangle, ceil() of the radius) -- the centre by ~0.2 mm and the scales by <= 0.3 %.  Returns the
    number of rounds.  The scene stays what it was for every practical purpose, but fp32 and fp64 can no longer disagree about
    WHICH pairs contribute."""
    g = torch.Generator().manual_seed(17)
    for rnd in range(max_rounds):
        d = O.knife_edge_gaussians(inp["means3D"], inp["opacities"], shs=inp.get("shs"), colors_precomp=inp.get("colors_precomp"),
                                   scales=inp.get("scales"), rotations=inp.get("rotations"), cov3D_precomp=inp.get("cov3D_precomp"),
                                   settings=s, detail=True)
        k, kg = d["alpha"], d["geometric"]
        if k.numel() == 0 and kg.numel() == 0:
            return rnd
        if k.numel():
            f = 1.0 + (0.002 + 0.018 * torch.rand(k.numel(), generator=g, dtype=torch.float64))
            o = inp["opacities"].clone()
            o[k, 0] = torch.where(o[k, 0] * f < 0.9985, o[k, 0] * f, o[k, 0] / f)
            inp["opacities"] = o.float().double()          # stays fp32-exact
        if kg.numel():
            m = inp["means3D"].clone()
            m[kg] += 2e-4 * torch.randn(kg.numel(), 3, generator=g, dtype=torch.float64)
            inp["means3D"] = m.float().double()
            if inp.get("scales") is not None:
                sc = inp["scales"].clone()
                sc[kg] *= 1.0 + 0.003 * torch.rand(kg.numel(), 1, generator=g, dtype=torch.float64)
                inp["scales"] = sc.float().double()
    raise AssertionError("scene still has knife-edge pairs after %d rounds" % max_rounds)


def depth_tie_gaussians(inp, s, ulps=32, window=64):
    """Visible Gaussians that have a NEAR TIE in view-space depth (within `ulps` fp32 ulp) with another visible Gaussian whose
    16x16-tile rectangle overlaps theirs -- the only pairs whose order the rounding of the depth itself can decide."""
    pp = O.preprocess(inp["means3D"], None, inp["opacities"], inp.get("shs"), inp.get("colors_precomp"), inp.get("scales"),
                      inp.get("rotations"), inp.get("cov3D_precomp"), None, None, s)
    vis = torch.nonzero(pp.visible).flatten()
    d = pp.depth.detach()[vis].double()
    order = torch.argsort(d)
    ds, ids, rect = d[order], vis[order], pp.rect[vis][order]
    hit = torch.zeros(ds.numel(), dtype=torch.bool)
    tol = ulps * ds.abs() * 2.0 ** -23
    for k in range(1, window + 1):
        if k >= ds.numel():
            break
        near = (ds[k:] - ds[:-k]) <= tol[:-k]
        if not bool(near.any()):
            break                              # (sorted: no pair further apart in the order can be nearer in depth)
        a, b = rect[:-k], rect[k:]
        overlap = (a[:, 0] < b[:, 2]) & (b[:, 0] < a[:, 2]) & (a[:, 1] < b[:, 3]) & (b[:, 1] < a[:, 3])
        m = near & overlap
        hit[:-k] |= m
        hit[k:] |= m
    else:
        raise AssertionError("depth ties: window too small")
    return ids[hit]


def move_off_knife_edges_and_depth_ties(inp, s, max_rounds=30, seed=23):
    """move_off_knife_edges + pushes overlapping near-ties in depth apart along the viewing direction (by 0.02-0.2 mm) until neither
    is left.  Returns (rounds, Gaussian moves).  After it, ANY two correct implementations order every tile's splats alike."""
    g = torch.Generator().manual_seed(seed)
    direction = s.viewmatrix.double().t()[2, :3].clone()
    moves = 0
    for rnd in range(max_rounds):
        rounds = move_off_knife_edges(inp, s)
        ties = depth_tie_gaussians(inp, s)
        if ties.numel() == 0 and rounds == 0:
            return rnd, moves
        if ties.numel() == 0:
            continue
        moves += int(ties.numel())
        m = inp["means3D"].clone()
        step = (2e-5 + 2e-4 * torch.rand(ties.numel(), 1, generator=g, dtype=torch.float64))
        sign = torch.where(torch.rand(ties.numel(), 1, generator=g) < 0.5, -1.0, 1.0).double()
        m[ties] += direction[None] * step * sign
        inp["means3D"] = m.float().double()
    raise AssertionError("near ties / knife edges did not clear after %d rounds" % max_rounds)


PER_GAUSSIAN = ("means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp")


def drop_knife_edges_and_depth_ties(inp, s, max_rounds=60):
    """The scene WITHOUT the Gaussians that sit on a cut-off or in a near tie of depth.  move_off_knife_edges() nudges such Gaussians,
    which converges only while a nudge is unlikely to land on the next knife edge -- splats of a few pixels.  A splat that covers
    hundreds of pixels has some pixel within the oracle's band of alpha = 1/255 at almost any opacity (tests/test_gpu_parity.py's
    `bg`, `wide`, `heavy`, ... never clear), so those scenes are thinned instead: whether a pair is near the alpha cut-off does not
    depend on the other Gaussians, and the transmittance cut-offs that do are rare -- a few rounds.  What remains is the same kind of
    scene (large splats, long lists) on which fp32 and fp64, or any two correct implementations, decide every cut-off alike.
    Returns (inputs, Gaussians kept, rounds)."""
    inp = dict(inp)
    n0 = inp["means3D"].shape[0]
    for rnd in range(max_rounds):
        d = O.knife_edge_gaussians(inp["means3D"], inp["opacities"], shs=inp.get("shs"), colors_precomp=inp.get("colors_precomp"),
                                   scales=inp.get("scales"), rotations=inp.get("rotations"), cov3D_precomp=inp.get("cov3D_precomp"),
                                   settings=s, detail=True)
        bad = torch.zeros(inp["means3D"].shape[0], dtype=torch.bool)
        bad[d["alpha"]] = True
        bad[d["geometric"]] = True
        bad[depth_tie_gaussians(inp, s)] = True
        if not bool(bad.any()):
            return inp, int(bad.numel()), rnd
        keep = ~bad
        for k in PER_GAUSSIAN:
            if inp.get(k) is not None:
                inp[k] = inp[k][keep].contiguous()
    raise AssertionError("knife edges / depth ties did not clear after %d rounds (%d of %d Gaussians left)" % (max_rounds, inp["means3D"].shape[0], n0))
