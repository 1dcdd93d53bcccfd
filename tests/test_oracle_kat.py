"""Closed-form known-answer tests for the oracle."""
import math

import pytest
import torch

from oracle import raster_oracle as O

DT = torch.float64
W, H, FX, FY, CX, CY = 64, 48, 50.0, 50.0, 32.0, 24.0   # cx = W/2 -> splat centre u = fx X/Z + cx - 0.5


def _render(xyz, scale, opac, rgb, bg=(0.0, 0.0, 0.0), w2c=None):
    n = len(xyz)
    s = O.make_settings(torch.eye(4, dtype=DT) if w2c is None else w2c, FX, FY, CX, CY, W, H,
                        bg=torch.tensor(bg, dtype=DT), dtype=DT)
    sh = ((torch.tensor(rgb, dtype=DT) - 0.5) / O.SH_C0).view(n, 1, 3)
    return O.rasterize(torch.tensor(xyz, dtype=DT), torch.zeros(n, 3, dtype=DT),
                       torch.tensor(opac, dtype=DT).view(n, 1), shs=sh,
                       scales=torch.tensor(scale, dtype=DT).view(n, 1).repeat(1, 3),
                       rotations=torch.tensor([[1.0, 0, 0, 0]] * n, dtype=DT), settings=s)


def _alpha(o, d2, sig2):
    a = min(0.99, o * math.exp(-0.5 * d2 / sig2))
    return a if a >= 1.0 / 255.0 else 0.0


def test_single_gaussian_on_axis():
    z, sc, o, c = 2.0, 0.05, 0.8, (0.2, 0.5, 0.9)
    col, radii, dep, opa, nt = _render([[0.0, 0.0, z]], [sc], [o], [c])
    sig2 = (FX * sc / z) ** 2 + 0.3
    assert radii.item() == math.ceil(3 * math.sqrt(sig2))
    gx, gy = CX - 0.5, CY - 0.5
    expect_touched = 0
    for py in range(H):
        for px in range(W):
            # tile-rect membership: 16x16 tiles overlapping [g-r, g+r]
            r = radii.item()
            tx0, tx1 = int((gx - r) / 16), int((gx + r + 15) / 16)
            ty0, ty1 = int((gy - r) / 16), int((gy + r + 15) / 16)
            inside = tx0 <= px // 16 < tx1 and ty0 <= py // 16 < ty1
            a = _alpha(o, (gx - px) ** 2 + (gy - py) ** 2, sig2) if inside else 0.0
            assert abs(opa[0, py, px].item() - a) < 1e-12
            assert abs(dep[0, py, px].item() - z * a) < 1e-12
            for ch in range(3):
                assert abs(col[ch, py, px].item() - c[ch] * a) < 1e-12
            if a > 0 and (1 - a) > 0.5:
                expect_touched += 1
    assert nt.item() == expect_touched


def test_two_coaxial_gaussians_sorted_by_depth():
    c1, c2 = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0)
    for order in ([1.5, 3.0], [3.0, 1.5]):
        col, radii, dep, opa, nt = _render([[0, 0, order[0]], [0, 0, order[1]]], [0.1, 0.1], [0.6, 0.7], [c1, c2])
        px, py = 31, 23        # d = (0.5, 0.5)
        zs = sorted(range(2), key=lambda i: order[i])
        o = [0.6, 0.7]
        a = [_alpha(o[i], 0.5, (FX * 0.1 / order[i]) ** 2 + 0.3) for i in range(2)]
        f, b = zs
        cols = [c1, c2]
        for ch in range(3):
            e = cols[f][ch] * a[f] + cols[b][ch] * a[b] * (1 - a[f])
            assert abs(col[ch, py, px].item() - e) < 1e-12
        assert abs(dep[0, py, px].item() - (order[f] * a[f] + order[b] * a[b] * (1 - a[f]))) < 1e-12
        assert abs(opa[0, py, px].item() - (1 - (1 - a[f]) * (1 - a[b]))) < 1e-12


def test_near_plane_is_patched_constant():
    col, radii, *_ = _render([[0, 0, 0.0011], [0, 0, 0.0009]], [1e-5, 1e-5], [0.5, 0.5], [(1, 1, 1), (1, 1, 1)])
    assert radii[0].item() > 0 and radii[1].item() == 0


def test_alpha_cutoff_and_transmittance_termination():
    # alpha just below / above 1/255 at the exact centre pixel (d = 0 -> G = 1)
    eps = 1e-6
    wc = None
    for o, vis in ((1 / 255 - eps, False), (1 / 255 + eps, True)):
        col, radii, dep, opa, nt = _render([[0.01 * 2 / FX * 0.0 + (0.5 / FX) * 2.0, (0.5 / FX) * 2.0, 2.0]], [0.05], [o], [(1, 1, 1)])
        assert (opa[0, 24, 32].item() > 0) == vis
    # stack of alpha=0.99 splats: T = 0.01, 1e-4 (composited: 1e-4 is not < 1e-4 in exact arithmetic but
    # 0.01*0.01 rounds below in binary), so probe with 0.9: T: .1, .01, .001, 1e-4(+), next would be 1e-5 -> excluded
    n = 8
    xyz = [[(0.5 / FX) * 2.0, (0.5 / FX) * 2.0, 2.0 + 0.1 * i] for i in range(n)]
    col, radii, dep, opa, nt = _render(xyz, [0.3] * n, [0.9] * n, [(1, 1, 1)] * n)
    T = 1.0
    for i in range(n):
        z = 2.0 + 0.1 * i
        G = math.exp(-0.5 * (((0.5 / FX) * 2.0 * FX / z + CX - 0.5 - 32) ** 2 * 2) / ((FX * 0.3 / z) ** 2 + 0.3))
        a = min(0.99, 0.9 * G)
        if T * (1 - a) < 1e-4:
            break
        T *= (1 - a)
    assert i < n - 1          # termination really happened inside the stack
    assert abs(opa[0, 24, 32].item() - (1 - T)) < 1e-9   # (1e-7 in the perspective divide)


def test_background_only_in_colour():
    bg = (0.2, 0.4, 0.6)
    col, radii, dep, opa, nt = _render([[0, 0, 2.0]], [0.05], [0.5], [(1.0, 1.0, 1.0)], bg=bg)
    a = opa[0, 23, 31].item()
    for ch in range(3):
        assert abs(col[ch, 23, 31].item() - (a + (1 - a) * bg[ch])) < 1e-12
        assert abs(col[ch, 0, 0].item() - bg[ch]) < 1e-12
    assert dep[0, 0, 0].item() == 0 and opa[0, 0, 0].item() == 0


def test_tile_coverage_at_tile_corner():
    # centre exactly on the corner shared by tiles (1,1),(2,1),(1,2),(2,2): pixel-space (31.5+.5, ...) -> u = 32 -> X = .5*z/fx
    z = 2.0
    X = (32.0 - (CX - 0.5)) * z / FX
    Y = (32.0 - (CY - 0.5)) * z / FY
    s = O.make_settings(torch.eye(4, dtype=DT), FX, FY, CX, CY, W, H, dtype=DT)
    pp = O.preprocess(torch.tensor([[X, Y, z]], dtype=DT), None, torch.tensor([[0.5]], dtype=DT),
                      torch.zeros(1, 1, 3, dtype=DT), None, torch.full((1, 3), 0.02, dtype=DT),
                      torch.tensor([[1.0, 0, 0, 0]], dtype=DT), None, None, None, s)
    assert torch.allclose(pp.xy, torch.tensor([[32.0, 32.0]], dtype=DT))
    x0, y0, x1, y1 = pp.rect[0].tolist()
    assert (x1 - x0) * (y1 - y0) == 4 and (x0, y0) == (1, 1)
    assert bool(pp.visible[0]) and pp.radii[0].item() == math.ceil(3 * math.sqrt((FX * 0.02 / z) ** 2 + 0.3))


def test_v_equals_zero_gives_background():
    col, radii, dep, opa, nt = _render([[0, 0, -1.0]], [0.1], [0.5], [(1, 1, 1)], bg=(0.1, 0.2, 0.3))
    assert radii.item() == 0 and nt.item() == 0
    assert torch.allclose(col[:, 5, 5], torch.tensor([0.1, 0.2, 0.3], dtype=DT))
