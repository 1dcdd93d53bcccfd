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
