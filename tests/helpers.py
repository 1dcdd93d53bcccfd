"""Shared scene builders for the parity tests (seeded, tiny; oracle finishes in seconds)."""
import math

import torch

from oracle import raster_oracle as O


def look_at_w2c(eye, target, up=(0.0, -1.0, 0.0), dtype=torch.float64):
    eye = torch.tensor(eye, dtype=dtype)
    target = torch.tensor(target, dtype=dtype)
    up = torch.tensor(up, dtype=dtype)
    z = target - eye
    z = z / z.norm()
    x = torch.linalg.cross(up, z)
    x = x / x.norm()
    y = torch.linalg.cross(z, x)
    R = torch.stack([x, y, z])          # rows = camera axes in world coords
    w2c = torch.eye(4, dtype=dtype)
    w2c[:3, :3] = R
    w2c[:3, 3] = -R @ eye
    return w2c


def random_scene(n, seed=0, dtype=torch.float64, W=64, H=48, fx=55.0, fy=52.0, cx=30.7, cy=24.9,
                 depth_range=(1.0, 4.0), scale_range=(0.02, 0.25), sh_degree=0, spread=1.2, bg=None,
                 pose_seed=None):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    # camera: small random rotation/translation away from identity
    ang = 0.2 * rn(3)
    Wm = O._hat(ang)
    Rm = torch.linalg.matrix_exp(Wm)
    w2c = torch.eye(4, dtype=torch.float64)
    w2c[:3, :3] = Rm
    w2c[:3, 3] = 0.3 * rn(3)
    # points in camera frame, then to world
    z = depth_range[0] + (depth_range[1] - depth_range[0]) * r(n)
    x = (r(n) - 0.5) * 2 * spread * z * (W / (2 * fx))
    y = (r(n) - 0.5) * 2 * spread * z * (H / (2 * fy))
    pc = torch.stack([x, y, z], dim=1)
    c2w = torch.linalg.inv(w2c)
    pw = pc @ c2w[:3, :3].t() + c2w[:3, 3]
    scales = scale_range[0] + (scale_range[1] - scale_range[0]) * r(n, 3) ** 2
    rot = rn(n, 4)
    rot = rot / rot.norm(dim=1, keepdim=True)
    opac = 0.05 + 0.94 * r(n, 1)
    K = (sh_degree + 1) ** 2
    shs = 0.8 * rn(n, K, 3)
    shs[:, 0] = (r(n, 3) - 0.5) / O.SH_C0
    if bg is None:
        bg = torch.zeros(3, dtype=torch.float64)
    s = O.make_settings(w2c, fx, fy, cx, cy, W, H, bg=bg, sh_degree=sh_degree, dtype=dtype)
    inp = dict(means3D=pw.to(dtype), means2D=torch.zeros(n, 3, dtype=dtype), opacities=opac.to(dtype),
               shs=shs.to(dtype), scales=scales.to(dtype), rotations=rot.to(dtype),
               theta=torch.zeros(3, dtype=dtype), rho=torch.zeros(3, dtype=dtype))
    return inp, s


def clone_inputs(inp, requires_grad=True, dtype=None):
    out = {}
    for k, v in inp.items():
        v = v.detach().clone()
        if dtype is not None:
            v = v.to(dtype)
        out[k] = v.requires_grad_(requires_grad)
    return out


def settings_to(s, dtype):
    return s._replace(bg=s.bg.to(dtype), viewmatrix=s.viewmatrix.to(dtype), projmatrix=s.projmatrix.to(dtype),
                      projmatrix_raw=s.projmatrix_raw.to(dtype), campos=s.campos.to(dtype))


def rel_err(a, b, floor=0.0):
    a = a.double()
    b = b.double()
    return ((a - b).abs().max() / (b.abs().max() + floor + 1e-300)).item()
