"""fp64 finite-difference pins for the oracle's autograd gradients."""
import pytest
import torch

from helpers import clone_inputs, random_scene
from oracle import raster_oracle as O


def _loss(out, wc, wd):
    col, radii, dep, opa, nt = out
    return (col * wc).sum() + (dep * wd).sum()


@pytest.mark.parametrize("seed,n,deg,bg", [(1, 12, 0, 0.0), (2, 30, 0, 0.3), (3, 10, 2, 0.0), (4, 8, 3, 0.1)])
def test_autograd_matches_central_differences(seed, n, deg, bg):
    inp, s = random_scene(n, seed=seed, sh_degree=deg, W=32, H=32, fx=30.0, fy=28.0, cx=15.2, cy=16.4,
                          bg=torch.full((3,), bg, dtype=torch.float64))
    g = torch.Generator().manual_seed(100 + seed)
    wc = torch.randn(3, 32, 32, generator=g, dtype=torch.float64)
    wd = torch.randn(1, 32, 32, generator=g, dtype=torch.float64)
    x = clone_inputs(inp)
    _loss(O.rasterize(**x, settings=s), wc, wd).backward()
    eps = 1e-6
    checked = 0
    for k in ["means3D", "opacities", "shs", "scales", "rotations"]:
        for trial in range(3):
            d = torch.randn(x[k].shape, generator=g, dtype=torch.float64)
            if k == "opacities":
                d = d * 0.1
            def central(e):
                xp = clone_inputs(inp, requires_grad=False)
                xm = clone_inputs(inp, requires_grad=False)
                xp[k] = xp[k] + e * d
                xm[k] = xm[k] - e * d
                op, om = O.rasterize(**xp, settings=s), O.rasterize(**xm, settings=s)
                return (_loss(op, wc, wd) - _loss(om, wc, wd)) / (2 * e)

            fd, fd_small = central(eps), central(eps / 4)
            if abs(fd - fd_small) > 1e-4 * max(1.0, abs(fd)):
                continue     # a radius / alpha cut-off was crossed: genuine piecewise-constant jump
            checked += 1
            an = (x[k].grad * d).sum()
            assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)), (k, trial, fd.item(), an.item())
    assert checked >= 12


@pytest.mark.parametrize("seed", [1, 2])
def test_pose_gradient_is_left_perturbation(seed):
    """d/dtau of render(SE3_exp(tau) @ W2C) at tau=0 == (grad_rho, grad_theta)  (pose_utils.py:66-98)."""
    inp, s = random_scene(14, seed=seed, W=32, H=32, fx=30.0, fy=28.0, cx=15.2, cy=16.4)
    g = torch.Generator().manual_seed(7 + seed)
    wc = torch.randn(3, 32, 32, generator=g, dtype=torch.float64)
    wd = torch.randn(1, 32, 32, generator=g, dtype=torch.float64)
    x = clone_inputs(inp)
    _loss(O.rasterize(**x, settings=s), wc, wd).backward()
    an = torch.cat([x["rho"].grad, x["theta"].grad])

    def render_with_tau(tau):
        w2c = s.viewmatrix.t()
        # exact SE3 exponential (pose_utils.py:30-78 general branch)
        rho, th = tau[:3], tau[3:]
        Wm = O._hat(th)
        ang = th.norm()
        I = torch.eye(3, dtype=torch.float64)
        if ang < 1e-12:
            R, V = I + Wm, I + 0.5 * Wm
        else:
            R = I + torch.sin(ang) / ang * Wm + (1 - torch.cos(ang)) / ang ** 2 * (Wm @ Wm)
            V = I + (1 - torch.cos(ang)) / ang ** 2 * Wm + (ang - torch.sin(ang)) / ang ** 3 * (Wm @ Wm)
        E = torch.eye(4, dtype=torch.float64)
        E[:3, :3] = R
        E[:3, 3] = V @ rho
        new = E @ w2c
        P = s.projmatrix_raw
        view = new.t().contiguous()
        s2 = s._replace(viewmatrix=view, projmatrix=view @ P, campos=s.campos)
        y = clone_inputs(inp, requires_grad=False)
        return _loss(O.rasterize(**y, settings=s2), wc, wd)

    eps = 1e-6
    for i in range(6):
        e = torch.zeros(6, dtype=torch.float64)
        e[i] = eps
        fd = (render_with_tau(e) - render_with_tau(-e)) / (2 * eps)
        assert abs(fd - an[i]) <= 2e-5 * max(1.0, abs(an[i])), (i, fd.item(), an[i].item())


def test_means2d_grad_is_ndc_scaled_pixel_gradient():
    """viewspace grad = dL/d(pixel centre) * (W/2, H/2): what densify_grad_threshold is calibrated on
    (gaussian_model.py:738-742)."""
    inp, s = random_scene(10, seed=9, W=32, H=32, fx=30.0, fy=28.0, cx=15.2, cy=16.4)
    x = clone_inputs(inp)
    pp = O.preprocess(x["means3D"], x["means2D"], x["opacities"], x["shs"], None, x["scales"], x["rotations"],
                      None, x["theta"], x["rho"], s)
    (pp.xy[:, 0].sum() * 3.0 + pp.xy[:, 1].sum() * 5.0).backward()
    assert torch.allclose(x["means2D"].grad[:, 0], torch.full((10,), 3.0 * 16.0, dtype=torch.float64))
    assert torch.allclose(x["means2D"].grad[:, 1], torch.full((10,), 5.0 * 16.0, dtype=torch.float64))
    assert torch.all(x["means2D"].grad[:, 2] == 0)


def test_upstream_pose_jacobian_switch_drops_exactly_the_principal_point_terms():
    """UPSTREAM_POSE_JACOBIAN: only (grad_theta, grad_rho) change, by the contraction of dL/d(ndc) with the
    dropped terms -P02/w, -P12/w; with a centred principal point the two forms coincide."""
    from helpers import random_scene
    from oracle import raster_oracle as O

    def grads(inp, s, flag):
        old = O.UPSTREAM_POSE_JACOBIAN
        O.UPSTREAM_POSE_JACOBIAN = flag
        try:
            x = {k: v.detach().clone().requires_grad_(True) for k, v in inp.items()}
            out = O.rasterize(x["means3D"], x["means2D"], x["opacities"], shs=x["shs"], scales=x["scales"], rotations=x["rotations"],
                              theta=x["theta"], rho=x["rho"], settings=s)
            g = torch.Generator().manual_seed(9)
            (out[0] * torch.randn(out[0].shape, generator=g, dtype=out[0].dtype)).sum().backward()
            return {k: v.grad.detach().clone() for k, v in x.items()}
        finally:
            O.UPSTREAM_POSE_JACOBIAN = old

    W, H = 64, 48
    inp, s = random_scene(60, seed=5, W=W, H=H, cx=W / 2, cy=H / 2)          # P02 = P12 = 0
    a, b = grads(inp, s, False), grads(inp, s, True)
    for k in a:
        assert torch.allclose(a[k], b[k], rtol=0, atol=1e-12 * max(1.0, a[k].abs().max().item())), k
    inp, s = random_scene(60, seed=5, W=W, H=H, cx=26.3, cy=29.9)              # off-centre principal point
    a, b = grads(inp, s, False), grads(inp, s, True)
    for k in a:
        if k not in ("theta", "rho"):
            assert torch.equal(a[k], b[k]), k
    P = s.projmatrix_raw.t()
    W2C = s.viewmatrix.t()
    pc = inp["means3D"] @ W2C[:3, :3].t() + W2C[:3, 3]
    g_ndc = a["means2D"][:, :2]
    dvz = -(g_ndc[:, 0] * P[0, 2] + g_ndc[:, 1] * P[1, 2]) / (pc[:, 2] + 1e-7)
    dv = torch.stack([torch.zeros_like(dvz), torch.zeros_like(dvz), dvz], dim=1)
    assert (b["rho"] - a["rho"] - dv.sum(0)).abs().max() < 1e-9 * max(1.0, a["rho"].abs().max().item())
    assert (b["theta"] - a["theta"] - torch.linalg.cross(pc, dv).sum(0)).abs().max() < 1e-9 * max(1.0, a["theta"].abs().max().item())
    assert (b["rho"] - a["rho"]).abs().max() > 1e-6 * a["rho"].abs().max()
