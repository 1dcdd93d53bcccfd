"""Closed-form known-answer tests for the oracle; the same cases run on the HIP path
in tests/test_gpu_kat.py."""
import math

import pytest
import torch

import kat_cases as K
from kat_cases import CX, CY, FX, FY, H, W
from oracle import raster_oracle as O

DT = torch.float64
TOL = 1e-12


def _render(xyz, scale, opac, rgb, bg=(0.0, 0.0, 0.0), w2c=None):
    n = len(xyz)
    s = O.make_settings(torch.eye(4, dtype=DT) if w2c is None else w2c, FX, FY, CX, CY, W, H,
                        bg=torch.tensor(bg, dtype=DT), dtype=DT)
    sh = ((torch.tensor(rgb, dtype=DT) - 0.5) / O.SH_C0).view(n, 1, 3)
    return O.rasterize(torch.tensor(xyz, dtype=DT), torch.zeros(n, 3, dtype=DT),
                       torch.tensor(opac, dtype=DT).view(n, 1), shs=sh,
                       scales=torch.tensor(scale, dtype=DT).view(n, 1).repeat(1, 3),
                       rotations=torch.tensor([[1.0, 0, 0, 0]] * n, dtype=DT), settings=s)


CASES = [K.single_gaussian_on_axis, K.two_coaxial_gaussians_sorted_by_depth, K.near_plane_is_patched_constant,
         K.alpha_cutoff_and_transmittance_termination, K.background_only_in_colour, K.tile_coverage_at_tile_corner,
         K.v_equals_zero_gives_background, K.n_touched_rule]


@pytest.mark.parametrize("case", CASES, ids=lambda f: f.__name__)
def test_known_answer(case):
    case(_render, TOL)


def test_tile_rectangle_at_tile_corner():
    # centre exactly on the corner shared by tiles (1,1),(2,1),(1,2),(2,2): u = 32 -> X = .5*z/fx
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


def _one(u, v, sc, opac=0.5, z=2.0):
    X, Y = (u - (CX - 0.5)) * z / FX, (v - (CY - 0.5)) * z / FY
    s = O.make_settings(torch.eye(4, dtype=DT), FX, FY, CX, CY, W, H, dtype=DT)
    return dict(means3D=torch.tensor([[X, Y, z]], dtype=DT), opacities=torch.tensor([[opac]], dtype=DT),
                shs=torch.zeros(1, 1, 3, dtype=DT), scales=torch.full((1, 3), sc, dtype=DT),
                rotations=torch.tensor([[1.0, 0, 0, 0]], dtype=DT)), s


def test_knife_edge_report_flags_the_integer_decisions_of_the_projection():
    """knife_edge_gaussians: a splat whose (centre - radius) sits within fp32 rounding of a multiple of 16 is reported (the
    tile rectangle trunc((xy -+ r)/16) may come out one tile wider in another implementation), one a quarter pixel away is
    not; likewise a radius 3 sqrt(lambda) within rounding of an integer."""
    sc, z = 0.02, 2.0
    r = math.ceil(3 * math.sqrt((FX * sc / z) ** 2 + 0.3 + math.sqrt(0.1)))   # 3 px (lambda = mid + sqrt(max(0.1, mid^2 - det)))
    for du, flagged in ((1e-5, True), (-1e-5, True), (0.25, False)):
        inp, s = _one(16.0 + r + du, 24.3, sc)                                # xy.x - r = 16 + du
        d = O.knife_edge_gaussians(inp["means3D"], inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                                   rotations=inp["rotations"], settings=s, detail=True)
        assert (d["geometric"].numel() == 1) == flagged, (du, d)
    # radius: choose the scale so that 3 sqrt(lambda) = 4 (1 + 1e-5) -> ceil() is a coin toss in fp32
    for rel, flagged in ((1e-5, True), (3e-2, False)):
        target = 4.0 * (1.0 + rel) / 3.0
        sc = math.sqrt(target * target - 0.3 - math.sqrt(0.1)) * z / FX
        inp, s = _one(30.37, 22.41, sc)
        d = O.knife_edge_gaussians(inp["means3D"], inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                                   rotations=inp["rotations"], settings=s, detail=True)
        assert (d["geometric"].numel() == 1) == flagged, (rel, d)


def test_knife_edge_report_flags_alpha_at_the_1_over_255_cutoff():
    """An isotropic splat centred ON a pixel whose opacity puts alpha at a chosen pixel exactly at 1/255 is reported; with the
    opacity 1 % higher it is not (every other pixel of a symmetric splat sits at a different, well separated alpha)."""
    sc, z = 0.06, 2.0
    sig2 = (FX * sc / z) ** 2 + 0.3
    G = math.exp(-0.5 * (3.0 ** 2) / sig2)                                   # the pixels 3 px to the left / right / above / below
    for opac, flagged in ((1.0 / 255.0 / G, True), (1.01 / 255.0 / G, False)):
        inp, s = _one(31.0, 23.0, sc, opac=opac)
        d = O.knife_edge_gaussians(inp["means3D"], inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                                   rotations=inp["rotations"], settings=s, detail=True)
        assert (d["alpha"].numel() == 1) == flagged, (opac, d)
