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


def run_hip(inp, s, wc=None, wd=None, dev="cuda:0"):
    """inp: dict of CPU tensors (any float dtype). Returns (outputs on CPU, grads dict on CPU or None)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    x = {k: v.detach().to(device=dev, dtype=torch.float32).requires_grad_(wc is not None) for k, v in inp.items()}
    rast = GaussianRasterizer(raster_settings=hip_settings(s, dev))
    out = rast(means3D=x["means3D"], means2D=x["means2D"], shs=x.get("shs"), colors_precomp=x.get("colors_precomp"),
               opacities=x["opacities"], scales=x.get("scales"), rotations=x.get("rotations"),
               cov3D_precomp=x.get("cov3D_precomp"), theta=x.get("theta"), rho=x.get("rho"))
    grads = None
    if wc is not None:
        loss = (out[0] * wc.to(dev).float()).sum() + (out[2] * wd.to(dev).float()).sum()
        loss.backward()
        grads = {k: (v.grad.detach().cpu() if v.grad is not None else None) for k, v in x.items()}
    torch.cuda.synchronize()
    return [o.detach().cpu() for o in out], grads


def run_oracle(inp, s, wc=None, wd=None, dtype=torch.float64):
    x = {k: v.detach().to(dtype).requires_grad_(wc is not None) for k, v in inp.items()}
    s2 = s._replace(bg=s.bg.to(dtype), viewmatrix=s.viewmatrix.to(dtype), projmatrix=s.projmatrix.to(dtype),
                    projmatrix_raw=s.projmatrix_raw.to(dtype), campos=s.campos.to(dtype))
    out = O.rasterize(x["means3D"], x["means2D"], x["opacities"], shs=x.get("shs"),
                      colors_precomp=x.get("colors_precomp"), scales=x.get("scales"), rotations=x.get("rotations"),
                      cov3D_precomp=x.get("cov3D_precomp"), theta=x.get("theta"), rho=x.get("rho"), settings=s2)
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
