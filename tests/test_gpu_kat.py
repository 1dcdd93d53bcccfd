"""-m gpu: the closed-form known-answer cases of SURVEY.md Appendix A.4 (items 1-6) on the HIP path, through the
drop-in Python surface and the C ABI: near plane 0.0011 / 0.0009 (/root/reference/README.md:88-92), alpha = 1/255 +- eps,
the T < 1e-4 terminator excluded, background only in colour, tile-corner coverage, the n_touched rule.  Same case bodies
as tests/test_oracle_kat.py (tests/kat_cases.py); fp32 tolerance."""
import pytest
import torch

import kat_cases as K
from kat_cases import CX, CY, FX, FY, H, W
from oracle import raster_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-6


def _render(xyz, scale, opac, rgb, bg=(0.0, 0.0, 0.0)):
    from diff_gaussian_rasterization import GaussianRasterizer
    from gpu_utils import hip_settings
    n = len(xyz)
    s = O.make_settings(torch.eye(4, dtype=torch.float64), FX, FY, CX, CY, W, H, bg=torch.tensor(bg, dtype=torch.float64),
                        dtype=torch.float64)
    f = lambda t: t.to(device=DEV, dtype=torch.float32).contiguous()
    sh = ((torch.tensor(rgb, dtype=torch.float64) - 0.5) / O.SH_C0).view(n, 1, 3)
    rast = GaussianRasterizer(raster_settings=hip_settings(s, DEV))
    out = rast(means3D=f(torch.tensor(xyz, dtype=torch.float64)), means2D=torch.zeros(n, 3, device=DEV),
               opacities=f(torch.tensor(opac, dtype=torch.float64).view(n, 1)), shs=f(sh),
               scales=f(torch.tensor(scale, dtype=torch.float64).view(n, 1).repeat(1, 3)),
               rotations=f(torch.tensor([[1.0, 0, 0, 0]] * n)))
    torch.cuda.synchronize()
    col, radii, dep, opa, nt = [o.detach().cpu() for o in out]
    return col.double(), radii, dep.double(), opa.double(), nt


CASES = [K.single_gaussian_on_axis, K.two_coaxial_gaussians_sorted_by_depth, K.near_plane_is_patched_constant,
         K.alpha_cutoff_and_transmittance_termination, K.background_only_in_colour, K.tile_coverage_at_tile_corner,
         K.v_equals_zero_gives_background, K.n_touched_rule]


@pytest.mark.parametrize("case", CASES, ids=lambda f: f.__name__)
def test_known_answer_on_hip(case):
    case(_render, TOL)
