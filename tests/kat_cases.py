"""Closed-form known-answer cases, shared by the oracle tests
(tests/test_oracle_kat.py: fp64 on the CPU) and the HIP tests (tests/test_gpu_kat.py: fp32 through the C ABI).

Every case takes  render(xyz, scale, opac, rgb, bg=(0, 0, 0)) -> (color, radii, depth, opacity, n_touched)  (CPU
tensors; identity pose, isotropic scales, identity rotations) and an absolute tolerance `tol` for image values.
Conventions that are citable: near plane 0.001 (/root/reference/README.md:88-92), consumers of opacity / n_touched
(/root/reference/thirdparty/monogs/utils/slam_utils.py:108-119, /root/reference/src/mapper.py:355,498,984).
"""
import math

W, H, FX, FY, CX, CY = 64, 48, 50.0, 50.0, 32.0, 24.0   # cx = W/2 -> splat centre u = fx X/Z + cx - 0.5


def alpha_of(o, d2, sig2):
    a = min(0.99, o * math.exp(-0.5 * d2 / sig2))
    return a if a >= 1.0 / 255.0 else 0.0


def single_gaussian_on_axis(render, tol):
    z, sc, o, c = 2.0, 0.05, 0.8, (0.2, 0.5, 0.9)
    col, radii, dep, opa, nt = render([[0.0, 0.0, z]], [sc], [o], [c])
    sig2 = (FX * sc / z) ** 2 + 0.3
    assert radii.item() == math.ceil(3 * math.sqrt(sig2))
    gx, gy = CX - 0.5, CY - 0.5
    r = radii.item()
    expect_touched = 0
    knife = 0
    for py in range(H):
        for px in range(W):
            # reachable pixels: the 16x16 reference tiles overlapping [g - r, g + r]
            tx0, tx1 = int((gx - r) / 16), int((gx + r + 15) / 16)
            ty0, ty1 = int((gy - r) / 16), int((gy + r + 15) / 16)
            inside = tx0 <= px // 16 < tx1 and ty0 <= py // 16 < ty1
            d2 = (gx - px) ** 2 + (gy - py) ** 2
            a = alpha_of(o, d2, sig2) if inside else 0.0
            raw = o * math.exp(-0.5 * d2 / sig2)
            if abs(raw - 1.0 / 255.0) < 10 * tol or abs((1 - a) - 0.5) < 10 * tol:
                knife += 1            # within rounding of a cut-off: either side is right at this tolerance
                continue
            assert abs(opa[0, py, px].item() - a) < tol
            assert abs(dep[0, py, px].item() - z * a) < 4 * tol
            for ch in range(3):
                assert abs(col[ch, py, px].item() - c[ch] * a) < tol
            if a > 0 and (1 - a) > 0.5:
                expect_touched += 1
    assert abs(nt.item() - expect_touched) <= knife


def two_coaxial_gaussians_sorted_by_depth(render, tol):
    c1, c2 = (1.0, 0.0, 0.0), (0.0, 1.0, 0.0)
    for order in ([1.5, 3.0], [3.0, 1.5]):
        col, radii, dep, opa, nt = render([[0, 0, order[0]], [0, 0, order[1]]], [0.1, 0.1], [0.6, 0.7], [c1, c2])
        px, py = 31, 23        # d = (0.5, 0.5)
        zs = sorted(range(2), key=lambda i: order[i])
        o = [0.6, 0.7]
        a = [alpha_of(o[i], 0.5, (FX * 0.1 / order[i]) ** 2 + 0.3) for i in range(2)]
        f, b = zs
        cols = [c1, c2]
        for ch in range(3):
            e = cols[f][ch] * a[f] + cols[b][ch] * a[b] * (1 - a[f])
            assert abs(col[ch, py, px].item() - e) < tol
        assert abs(dep[0, py, px].item() - (order[f] * a[f] + order[b] * a[b] * (1 - a[f]))) < 4 * tol
        assert abs(opa[0, py, px].item() - (1 - (1 - a[f]) * (1 - a[b]))) < tol


def near_plane_is_patched_constant(render, tol):
    col, radii, *_ = render([[0, 0, 0.0011], [0, 0, 0.0009]], [1e-5, 1e-5], [0.5, 0.5], [(1, 1, 1), (1, 1, 1)])
    assert radii[0].item() > 0 and radii[1].item() == 0


def alpha_cutoff_and_transmittance_termination(render, tol):
    # alpha just below / above 1/255 at the exact centre pixel (d = 0 -> G = 1)
    eps = 1e-6
    for o, vis in ((1 / 255 - eps, False), (1 / 255 + eps, True)):
        col, radii, dep, opa, nt = render([[(0.5 / FX) * 2.0, (0.5 / FX) * 2.0, 2.0]], [0.05], [o], [(1, 1, 1)])
        assert (opa[0, 24, 32].item() > 0) == vis
    # stack of opacity-0.9 splats: T = .1, .01, .001, 1e-4(+); the splat that would push T below 1e-4 is excluded
    n = 8
    xyz = [[(0.5 / FX) * 2.0, (0.5 / FX) * 2.0, 2.0 + 0.1 * i] for i in range(n)]
    col, radii, dep, opa, nt = render(xyz, [0.3] * n, [0.9] * n, [(1, 1, 1)] * n)
    T = 1.0
    for i in range(n):
        z = 2.0 + 0.1 * i
        G = math.exp(-0.5 * (((0.5 / FX) * 2.0 * FX / z + CX - 0.5 - 32) ** 2 * 2) / ((FX * 0.3 / z) ** 2 + 0.3))
        a = min(0.99, 0.9 * G)
        if T * (1 - a) < 1e-4:
            break
        T *= (1 - a)
    assert i < n - 1          # termination really happened inside the stack
    assert T * (1 - a) < 0.5e-4 and T > 1.001e-4, "the case must not sit on the cut-off itself"
    assert abs(opa[0, 24, 32].item() - (1 - T)) < max(tol, 1e-9)   # (1e-7 in the perspective divide)


def background_only_in_colour(render, tol):
    bg = (0.2, 0.4, 0.6)
    col, radii, dep, opa, nt = render([[0, 0, 2.0]], [0.05], [0.5], [(1.0, 1.0, 1.0)], bg=bg)
    a = opa[0, 23, 31].item()
    assert a > 0.1
    for ch in range(3):
        assert abs(col[ch, 23, 31].item() - (a + (1 - a) * bg[ch])) < tol
        assert abs(col[ch, 0, 0].item() - bg[ch]) < tol
    assert dep[0, 0, 0].item() == 0 and opa[0, 0, 0].item() == 0


def tile_coverage_at_tile_corner(render, tol):
    """A splat centred exactly on the corner shared by four 16x16 reference tiles reaches pixels of all four and of no
    other tile; radii and the visibility filter agree."""
    z, sc, o = 2.0, 0.02, 0.9
    X = (32.0 - (CX - 0.5)) * z / FX
    Y = (32.0 - (CY - 0.5)) * z / FY
    col, radii, dep, opa, nt = render([[X, Y, z]], [sc], [o], [(1, 1, 1)])
    sig2 = (FX * sc / z) ** 2 + 0.3
    assert radii.item() == math.ceil(3 * math.sqrt(sig2)) and radii.item() > 0
    for py, px in ((31, 31), (31, 32), (32, 31), (32, 32)):        # one pixel of each of the four tiles
        a = alpha_of(o, (32.0 - px) ** 2 + (32.0 - py) ** 2, sig2)
        # (off the optical axis the footprint is only approximately isotropic: J has a perspective column)
        assert a > 0.1 and abs(opa[0, py, px].item() - a) < 0.03 * a
    assert opa[0, :16].abs().max().item() == 0 and opa[0, :, :16].abs().max().item() == 0 and opa[0, :, 48:].abs().max().item() == 0


def v_equals_zero_gives_background(render, tol):
    col, radii, dep, opa, nt = render([[0, 0, -1.0]], [0.1], [0.5], [(1, 1, 1)], bg=(0.1, 0.2, 0.3))
    assert radii.item() == 0 and nt.item() == 0
    for ch, v in enumerate((0.1, 0.2, 0.3)):
        assert abs(col[ch, 5, 5].item() - v) < max(tol, 1e-7)
    assert dep.abs().max().item() == 0 and opa.abs().max().item() == 0


def n_touched_rule(render, tol):
    """n_touched counts the pixels a splat was composited at while the transmittance AFTER it stayed above 0.5 -- the
    quantity whose `> 0` test drives visibility sets (mapper.py:355,498,984).  Front splat with alpha ~0.6 everywhere
    near its centre: it is composited there but T' = 0.4 < 0.5, so only its faint rim counts; the splat behind it sees
    T <= 0.4 < 0.5 at those pixels and counts only outside the front splat's core."""
    z1, z2, sc = 2.0, 2.5, 0.2
    col, radii, dep, opa, nt = render([[0, 0, z1], [0, 0, z2]], [sc, sc], [0.6, 0.3], [(1, 0, 0), (0, 1, 0)])
    gx, gy = CX - 0.5, CY - 0.5
    s1, s2 = (FX * sc / z1) ** 2 + 0.3, (FX * sc / z2) ** 2 + 0.3
    e1 = e2 = knife = 0
    for py in range(H):
        for px in range(W):
            d2 = (gx - px) ** 2 + (gy - py) ** 2
            a1, a2 = alpha_of(0.6, d2, s1), alpha_of(0.3, d2, s2)
            if min(abs((1 - a1) - 0.5), abs((1 - a1) * (1 - a2) - 0.5)) < 1e-4 or \
               min(abs(0.6 * math.exp(-0.5 * d2 / s1) - 1 / 255), abs(0.3 * math.exp(-0.5 * d2 / s2) - 1 / 255)) < 1e-5:
                knife += 1
                continue
            e1 += 1 if (a1 > 0 and (1 - a1) > 0.5) else 0
            e2 += 1 if (a2 > 0 and (1 - a1) * (1 - a2) > 0.5) else 0
    assert e1 > 50 and e2 > 50
    assert abs(nt[0].item() - e1) <= knife and abs(nt[1].item() - e2) <= knife
