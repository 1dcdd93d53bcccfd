"""Host-side GaussianModel pieces that need no GPU: PLY checkpoint format, seeding from an RGB-D frame."""
import numpy as np
import torch

from splat_slam_amd.gaussian_model import GaussianModel, OptParams


def _model(n=37, deg=0, seed=0):
    g = torch.Generator().manual_seed(seed)
    gm = GaussianModel(deg, config=None, device="cpu")
    gm.training_setup(OptParams())
    K = (deg + 1) ** 2
    feats = torch.randn(n, 3, K, generator=g)
    gm.extend_from_pcd(torch.randn(n, 3, generator=g), feats, torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g),
                       torch.randn(n, 1, generator=g), 3)
    return gm


def test_ply_roundtrip_and_layout(tmp_path):
    for deg in (0, 1):
        gm = _model(deg=deg)
        path = str(tmp_path / f"map{deg}.ply")
        gm.save_ply(path)
        raw = open(path, "rb").read()
        head, body = raw.split(b"end_header\n", 1)
        lines = head.decode().strip().split("\n")
        assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
        props = [l.split()[-1] for l in lines[3:]]
        K = (deg + 1) ** 2
        expect = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(3 * (K - 1))]
                  + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])
        assert props == expect                       # gaussian_model.py:331-343
        assert all(l.split()[1] == "float" for l in lines[3:])
        assert len(body) == 37 * len(expect) * 4
        arr = np.frombuffer(body, dtype="<f4").reshape(37, -1)
        assert np.array_equal(arr[:, :3], gm._xyz.detach().numpy()) and np.all(arr[:, 3:6] == 0)
        # f_dc / f_rest are stored channel-major (transpose(1,2).flatten), like the reference
        assert np.array_equal(arr[:, 6:9], gm._features_dc.detach().transpose(1, 2).flatten(1).numpy())
        other = GaussianModel(deg, device="cpu")
        other.load_ply(path)
        for name in ["_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"]:
            assert torch.equal(getattr(other, name).detach(), getattr(gm, name).detach()), name
        assert other.active_sh_degree == deg and other.max_radii2D.shape == (37,)


def test_seeding_from_rgbd_backprojects_valid_pixels_only():
    cfg = {"mapping": {"pcd_downsample": 4, "pcd_downsample_init": 2, "adaptive_pointsize": True, "point_size": 0.05}}
    gm = GaussianModel(0, config=cfg, device="cpu", knn_fn=lambda p: torch.full((p.shape[0],), 0.04))
    gm.training_setup(OptParams())

    class Cam:
        pass
    cam = Cam()
    H, W = 12, 16
    cam.fx = cam.fy = 10.0
    cam.cx, cam.cy = 7.5, 5.5
    cam.R, cam.T = torch.eye(3), torch.tensor([0.1, -0.2, 0.3])
    cam.exposure_a, cam.exposure_b = torch.zeros(1), torch.zeros(1)
    cam.original_image = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1))
    depth = torch.full((H, W), 2.0)
    depth[:3] = 0.0                                   # invalid rows are never seeded
    torch.manual_seed(0)
    gm.extend_from_pcd_seq(cam, kf_id=5, init=True, depthmap=depth)
    n_valid = int((depth > 0).sum())
    assert gm.get_xyz.shape[0] == n_valid // 2        # pcd_downsample_init
    pc = gm.get_xyz.detach() @ cam.R.T + cam.T        # back in the camera frame: every point sits at z = 2 on a valid pixel
    assert torch.allclose(pc[:, 2], torch.full((pc.shape[0],), 2.0), atol=1e-5)
    v = pc[:, 1] * cam.fy / pc[:, 2] + cam.cy
    assert float(v.min()) > 2.5
    # scale = log sqrt(clamp(knn, 1e-7) * point_size), quaternion identity, opacity logit(0.5) (gaussian_model.py:194-215)
    assert torch.allclose(gm._scaling.detach(), torch.full_like(gm._scaling, float(np.log(np.sqrt(0.04 * 0.05)))), atol=1e-6)
    assert torch.equal(gm._rotation.detach()[:, 0], torch.ones(pc.shape[0])) and gm._rotation.detach()[:, 1:].abs().max() == 0
    assert gm._opacity.detach().abs().max() < 1e-6 and torch.all(gm.unique_kfIDs == 5)


def test_pair_estimates_carried_across_map_sizes():
    """FusedMappingLoop sizes a camera's workspace from carried-over measurements instead of a probe render per map size."""
    from splat_slam_amd.fused import estimate_pairs
    assert estimate_pairs({}, 3, 1000) is None                                   # nothing known: the caller probes
    hints = {3: (40000, 100000), 7: (90000, 200000)}
    assert estimate_pairs(hints, 3, 100000) == int(40000 * 1.25) + 1024           # own measurement, same map size
    assert estimate_pairs(hints, 3, 150000) == int(40000 * 1.5 * 1.25) + 1024     # scaled with the growth of the map
    assert estimate_pairs(hints, 3, 50000) == int(40000 * 1.25) + 1024            # never scaled down
    new_cam = estimate_pairs(hints, 11, 200000)                                   # never measured: 1.5 x the largest of the others
    assert new_cam == int(1.5 * max(int(40000 * 2.0 * 1.25) + 1024, int(90000 * 1.25) + 1024))


def test_tile_sort_network_index_arithmetic():
    """csrc/sgr_blend.hip wave_sort_any: the bitonic 'mirror' network (every compare-exchange ascending, virtual +inf padding
    behind n never moves) with the pair index computed by shifts -- i = ((t >> lj) << (lj + 1)) + (t & (j - 1)), j = 1 << lj --
    restated here: it must be the same index as the division form and must sort ANY n."""
    import random
    rnd = random.Random(7)
    for n in list(range(0, 70)) + [127, 128, 129, 200, 255, 256, 257, 300, 1000]:
        a = [rnd.getrandbits(40) for _ in range(n)]
        want = sorted(a)
        P = 1
        while P < n:
            P <<= 1
        k, lk = 2, 1
        while k <= P:
            j, lj = k >> 1, lk - 1
            while j > 0:
                first = j == (k >> 1)
                for t in range(P >> 1):
                    i = ((t >> lj) << (lj + 1)) + (t & (j - 1))
                    assert i == (t // j) * (j << 1) + (t % j)
                    l = (i ^ (k - 1)) if first else (i ^ j)
                    if l < i:
                        i, l = l, i
                    if l < n and a[l] < a[i]:
                        a[i], a[l] = a[l], a[i]
                j, lj = j >> 1, lj - 1
            k, lk = k << 1, lk + 1
        assert a == want, n


def test_register_blocked_tile_sort_network():
    """csrc/sgr_blend.hip wave_sort_registers<KPL>: 64 lanes x KPL keys, element e = lane * KPL + r.  Stage (KK, J): J < KPL
    compares registers r and r | J of one lane, J >= KPL exchanges register r with lane ^ (J / KPL); a block of KK elements is
    descending iff (e & KK) != 0; ONE compare per exchange, the direction enters as an xor (keys are unique up to the ~0
    padding).  Restated here lane by lane: it must sort any count <= 64 * KPL and push the padding to the end."""
    import random
    rnd = random.Random(11)
    PAD = (1 << 64) - 1
    for KPL in (2, 4, 8):
        for count in [65, 64 * KPL - 1, 64 * KPL, 64 * KPL // 2 + 3, rnd.randrange(65, 64 * KPL)]:
            vals = [rnd.getrandbits(63) for _ in range(count)]
            k = [[vals[l * KPL + r] if l * KPL + r < count else PAD for r in range(KPL)] for l in range(64)]
            KK = 2
            while KK <= 64 * KPL:
                J = KK >> 1
                while J >= 1:
                    if J < KPL:
                        for lane in range(64):
                            for r in range(KPL):
                                if r & J:
                                    continue
                                down = ((r & KK) != 0) if KK < KPL else (((lane * KPL) & KK) != 0)
                                a, b = k[lane][r], k[lane][r | J]
                                if (b < a) != down:
                                    k[lane][r], k[lane][r | J] = b, a
                    else:
                        M = J // KPL
                        new = [row[:] for row in k]
                        for lane in range(64):
                            lower, up = (lane & M) == 0, ((lane * KPL) & KK) == 0
                            keep_max = lower != up
                            for r in range(KPL):
                                mine, other = k[lane][r], k[lane ^ M][r]
                                if (other < mine) != keep_max:
                                    new[lane][r] = other
                        k = new
                    J >>= 1
                KK <<= 1
            flat = [k[l][r] for l in range(64) for r in range(KPL)]
            assert flat[:count] == sorted(vals) and all(x == PAD for x in flat[count:]), (KPL, count)


def test_backward_chunk_plan_covers_every_list_position_once():
    """csrc/sgr_blend.hip tile_backward: a list of `eff` splats is cut from the far end into chunks of 64 / 32 / 16 / 8 / 4 lanes --
    the binary expansion of the length, a width rounded up only from 13 / 29 / 61 splats (round 4; round 3 rounded up from 3/4 full).
    Every position belongs to exactly one chunk, chunks run back to front, and the plan never takes more loop iterations (32 * width
    / 64 per chunk) than round 3's."""
    def plan(eff, cut):
        # instructions of one loop iteration by chunk width (ISA of the round-4 build, un-stashed loops) and of a chunk's prologue + epilogue
        per_iteration, per_chunk = {64: 85, 32: 83, 16: 76, 8: 69, 4: 71}, 95
        end, seen, iters, cost = eff, [], 0, 0
        while end > 0:
            gw = 64 if end >= cut[0] else 32 if end >= cut[1] else 16 if end >= cut[2] else 8 if end >= 5 else 4
            start = 0 if gw >= end else end - gw
            assert end - start <= gw
            seen = list(range(start, end)) + seen
            iters += 32 * gw // 64
            cost += (32 * gw // 64) * per_iteration[gw] + per_chunk
            end = start
        assert seen == list(range(eff))
        return iters, cost

    for eff in range(1, 700):
        iters, cost = plan(eff, (61, 29, 13))
        old_iters, old_cost = plan(eff, (48, 24, 12))
        assert iters <= old_iters, (eff, iters, old_iters)
        assert cost <= old_cost, (eff, cost, old_cost)        # (fewer iterations always pay for the extra chunk prologues)


def test_narrow_backward_groups_interleaved_in_their_dpp_row():
    """csrc/sgr_blend.hip bwd_chunk2: for GW = 8 / 4 the 16 / GW groups of a DPP row own the lanes l16 % (16 / GW); a lane's position in
    its group is l16 / (16 / GW).  (lane -> (group, position)) must be a bijection onto 64 / GW groups x GW positions, "the previous
    lane of my group" must be the lane 16 / GW below IN THE SAME ROW (what row_shr:(16/GW) reads) and must not exist for position 0
    (the DPP source is then out of the row: the lane keeps its value), and the quad_perm pairing of the chunk's final reduction must
    meet all groups of a row."""
    for GW in (8, 4):
        gpr = 16 // GW
        seen = {}
        for lane in range(64):
            row, l16 = lane >> 4, lane & 15
            sub, sl = row * gpr + (l16 & (gpr - 1)), l16 // gpr
            assert 0 <= sub < 64 // GW and 0 <= sl < GW
            assert (sub, sl) not in seen
            seen[(sub, sl)] = lane
        assert len(seen) == 64
        for (sub, sl), lane in seen.items():
            src = (lane & 15) - gpr                         # row_shr:gpr
            if sl == 0:
                assert src < 0                              # out of the row: the lane is disabled, keeps its value
            else:
                assert src >= 0 and seen[(sub, sl - 1)] == (lane & ~15) + src
        # final reduction: lanes l ^ 1 (GW = 8) or l ^ 2 then l ^ 1 (GW = 4) hold the same position of the row's other groups
        for lane in range(64):
            row, l16 = lane >> 4, lane & 15
            partners = {lane, lane ^ 1} if GW == 8 else {lane, lane ^ 1, lane ^ 2, lane ^ 3}
            assert {((p & 15) // gpr) for p in partners} == {l16 // gpr}
            assert {row * gpr + ((p & 15) & (gpr - 1)) for p in partners} == set(range(row * gpr, row * gpr + gpr))


def test_tile_of_block_is_a_bijection_and_its_division_is_exact():
    """csrc/sgr_blend.hip tile_of_block: workgroup b (hardware places it on XCD b % 8) -> 8x8 tile.  Every XCD walks a contiguous run of
    16x16 super tiles, four tiles each; every tile of the image is owned by exactly one block of the grid 8 * 4 * ceil(nsuper / 8);
    the division by the super-tile row length is a multiply-high by floor(2^32 / sgx) + 1 (LOff.sgx_magic), exact while
    st * sgx < 2^32."""
    for W, H in ((640, 480), (640, 320), (512, 384), (320, 240), (96, 64), (50, 37), (16, 16), (1296, 968), (4000, 3000)):
        sgx, sgy = (W + 15) // 16, (H + 15) // 16
        gx, gy = (W + 7) // 8, (H + 7) // 8
        nsuper = sgx * sgy
        per = (nsuper + 7) >> 3
        magic = (1 << 32) // sgx + 1
        assert nsuper * sgx < 1 << 32
        owners = {}
        for b in range(8 * 4 * per):
            j = b >> 3
            st, wv = (b & 7) * per + (j >> 2), j & 3
            if st >= nsuper or j >= 4 * per:
                continue
            row = (st * magic) >> 32
            assert row == st // sgx
            tx, ty = (st - row * sgx) * 2 + (wv & 1), row * 2 + (wv >> 1)
            if tx >= gx or ty >= gy:
                continue
            assert (tx, ty) not in owners
            owners[(tx, ty)] = b
        assert len(owners) == gx * gy
        # the four tiles of a super tile sit on ONE XCD, in consecutive slots of its queue
        for (tx, ty), b in owners.items():
            assert owners.get((tx ^ 1, ty), b) & 7 == b & 7 and owners.get((tx, ty ^ 1), b) & 7 == b & 7


def test_g_stash_fits_the_unused_part_of_the_wave_slice():
    """csrc/sgr_blend.hip: a list of <= 16 splats is staged pair-interleaved (14 floats per splat = 112 bytes per pair, an odd list
    padded by one splat) in the first 1008 bytes of the wave's LDS slice; the stash (16 rows of 64 pixels, 66 floats apart) must end
    before the pixel state that follows the staging (64 x 56 B) and sorted-index (512 x 4 B) areas, and a row stride of 66 floats keeps
    the 16 lanes of a group on distinct bank pairs for their 8-byte reads."""
    staging = ((16 >> 1) + 1) * 112
    assert staging == 1008
    assert staging + 16 * 66 * 4 <= 64 * 56 + 512 * 4
    assert 512 * 4 + 64 * 56 + 2 * 64 * 16 == 7680 and 160 * 1024 // 7680 >= 20          # the slice still leaves 5 waves per SIMD resident
    for gp in range(32):
        banks = set()
        for idx in range(16):
            a = idx * 66 + 2 * gp
            assert a % 2 == 0                              # 8-byte aligned
            banks |= {a % 64, (a + 1) % 64}
        assert len(banks) == 32


def test_plane_cull_is_conservative_for_the_exact_rectangle_test():
    """csrc/sgr_preprocess.hip maybe_visible_planes (phase A of K1), restated in fp32 numpy next to the exact decision of
    preprocess_view (near plane + non-empty reference-tile rectangle): the cheap test may pass Gaussians the exact path rejects,
    never the other way round -- for splats of every size, in front of / beside / behind the camera, any principal point."""
    import numpy as np
    rng = np.random.default_rng(3)
    f32 = np.float32
    near = f32(0.001)
    n_vis = n_pass = 0
    for trial in range(40):
        W, H = [(640, 480), (640, 320), (96, 64), (321, 243)][trial % 4]
        fx, fy = f32(rng.uniform(0.5, 1.2) * W), f32(rng.uniform(0.5, 1.2) * W)
        cx, cy = f32(W / 2 + rng.uniform(-20, 20)), f32(H / 2 + rng.uniform(-20, 20))
        tanfovx, tanfovy = f32(W / (2 * fx)), f32(H / (2 * fy))
        # rigid W2C
        a = rng.normal(size=3)
        a /= np.linalg.norm(a)
        th = rng.uniform(0, np.pi)
        Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        t = rng.normal(size=3)
        w2c = np.eye(4)
        w2c[:3, :3], w2c[:3, 3] = R, t
        P = np.zeros((4, 4))
        zn, zf = 0.01, 100.0
        P[0, 0], P[1, 1] = 2 * fx / W, 2 * fy / H
        P[0, 2], P[1, 2] = 2 * (cx + 0.5) / W - 1.0, 2 * (cy + 0.5) / H - 1.0      # (principal-point offset: any value will do here)
        P[3, 2], P[2, 2], P[2, 3] = 1.0, zf / (zf - zn), -(zf * zn) / (zf - zn)
        vm = w2c.T.astype(f32).reshape(-1)                   # transposed layout: W2C[r][c] = vm[c*4+r]
        pm = (w2c.T @ P.T).astype(f32).reshape(-1)
        sgx, sgy = (W + 15) // 16, (H + 15) // 16
        n = 4000
        pc = np.concatenate([rng.normal(size=(n // 2, 3)) * [3, 3, 3] + [0, 0, 3], rng.normal(size=(n - n // 2, 3)) * 0.2 + [0, 0, 0.05]])
        pts = ((pc - t) @ R).astype(f32)                     # world points spread around the frustum
        scales = np.exp(rng.uniform(np.log(1e-3), np.log(0.8), size=(n, 3))).astype(f32)
        q = rng.normal(size=(n, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        r_, x, y, z = q.T
        Rq = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r_ * z), 2 * (x * z + r_ * y), 2 * (x * y + r_ * z), 1 - 2 * (x * x + z * z),
                       2 * (y * z - r_ * x), 2 * (x * z - r_ * y), 2 * (y * z + r_ * x), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)
        M = Rq * scales[:, None, :]
        Sig = (M @ M.transpose(0, 2, 1)).astype(f32)
        trS = f32(1.01) * (scales ** 2).sum(1)
        # ---- exact (preprocess_view)
        Wm = vm.reshape(4, 4).T[:3, :3]
        pv = pts @ Wm.T + vm.reshape(4, 4).T[:3, 3]
        front = pv[:, 2] > near
        ph = np.concatenate([pts, np.ones((n, 1), f32)], 1) @ pm.reshape(4, 4)
        pw = f32(1) / (ph[:, 3] + f32(1e-7))
        ndcx, ndcy = ph[:, 0] * pw, ph[:, 1] * pw
        limx, limy = f32(1.3) * tanfovx, f32(1.3) * tanfovy
        fxp, fyp = f32(W) / (2 * tanfovx), f32(H) / (2 * tanfovy)
        tz = np.where(front, pv[:, 2], 1)
        tx = np.clip(pv[:, 0] / tz, -limx, limx) * tz
        ty = np.clip(pv[:, 1] / tz, -limy, limy) * tz
        J = np.zeros((n, 2, 3), f32)
        J[:, 0, 0], J[:, 0, 2], J[:, 1, 1], J[:, 1, 2] = fxp / tz, -fxp * tx / tz ** 2, fyp / tz, -fyp * ty / tz ** 2
        T = J @ Wm
        cov = T @ Sig @ T.transpose(0, 2, 1)
        ca, cb, cc = cov[:, 0, 0] + f32(0.3), cov[:, 0, 1], cov[:, 1, 1] + f32(0.3)
        det = ca * cc - cb * cb
        mid = 0.5 * (ca + cc)
        lam = mid + np.sqrt(np.maximum(0.1, mid * mid - det))
        rad = np.ceil(3 * np.sqrt(lam))
        px, py = ((ndcx + 1) * W - 1) * 0.5, ((ndcy + 1) * H - 1) * 0.5
        tr = lambda v: np.trunc(v).astype(np.int64)
        rx0 = np.clip(tr((px - rad) / 16), 0, sgx)
        rx1 = np.clip(tr((px + rad + 15) / 16), 0, sgx)
        ry0 = np.clip(tr((py - rad) / 16), 0, sgy)
        ry1 = np.clip(tr((py + rad + 15) / 16), 0, sgy)
        with np.errstate(all="ignore"):
            exact = front & (det != 0) & np.isfinite(px) & np.isfinite(py) & np.isfinite(rad) & ((rx1 - rx0) * (ry1 - ry0) != 0)
        # ---- plane test (fp32, the kernel's operation order)
        G = np.abs(Wm.astype(f32) @ Wm.astype(f32).T).sum(1)
        K = f32(3) * np.sqrt(G.max() * (fxp * fxp * (1 + limx * limx) + fyp * fyp * (1 + limy * limy))) * f32(1.003)
        s = np.sqrt(np.maximum(trS, 0)).astype(f32) * f32(1.0005)
        zc = vm[2] * pts[:, 0] + vm[6] * pts[:, 1] + vm[10] * pts[:, 2] + vm[14]
        w = pm[3] * pts[:, 0] + pm[7] * pts[:, 1] + pm[11] * pts[:, 2] + pm[15] + f32(1e-7)
        ph0 = pm[0] * pts[:, 0] + pm[4] * pts[:, 1] + pm[8] * pts[:, 2] + pm[12]
        ph1 = pm[1] * pts[:, 0] + pm[5] * pts[:, 1] + pm[9] * pts[:, 2] + pm[13]
        hx, hy, ksw = f32(0.5 * W), f32(0.5 * H), K * s * w
        ax, ay = hx * ph0, hy * ph1
        l = (ax + (hx + f32(4.5)) * w) * zc + ksw
        r = (ax + (hx - f32(5.5) - f32(sgx * 16)) * w) * zc - ksw
        tt = (ay + (hy + f32(4.5)) * w) * zc + ksw
        b = (ay + (hy - f32(5.5) - f32(sgy * 16)) * w) * zc - ksw
        planes = (zc > near * f32(0.999)) & (~(w > 0) | (~(l < 0) & ~(r >= 0) & ~(tt < 0) & ~(b >= 0)))
        assert not bool((exact & ~planes).any()), f"trial {trial}: the plane test rejected {int((exact & ~planes).sum())} visible Gaussians"
        n_vis += int(exact.sum())
        n_pass += int(planes.sum())
    assert n_vis > 10000 and n_pass < 2.0 * n_vis, (n_vis, n_pass)       # conservative, yet a real filter


def test_footprint_bin_test_is_a_lower_bound_of_the_quadratic_form():
    """csrc/sgr_common.h footprint_qmin, restated: the minimum of q = A dx^2 + 2 B dx dy + C dy^2 over the bounding box of a
    bin's 8x8 pixel centres (0 if the centre is inside, else the smallest of the four edge minima).  It must never exceed q at
    any pixel of the bin -- so 'qmin > tau' proves that no pixel reaches alpha >= 1/255 -- and must be attained on the box."""
    import numpy as np
    rng = np.random.default_rng(5)

    def qmin(px, py, A, B, C, tx, ty):
        xl, yl = tx * 8 - px, ty * 8 - py
        xh, yh = xl + 7, yl + 7
        if xl <= 0 <= xh and yl <= 0 <= yh:
            return 0.0
        ex = lambda c: (lambda d: A * c * c + 2 * B * c * d + C * d * d)(min(yh, max(yl, -B * c / C)))
        ey = lambda c: (lambda d: A * d * d + 2 * B * d * c + C * c * c)(min(xh, max(xl, -B * c / A)))
        return min(ex(xl), ex(xh), ey(yl), ey(yh))

    for _ in range(300):
        s1, s2, th = rng.uniform(0.6, 30.0), rng.uniform(0.6, 30.0), rng.uniform(0, np.pi)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        conic = np.linalg.inv(R @ np.diag([s1 * s1, s2 * s2]) @ R.T)
        A, B, C = conic[0, 0], conic[0, 1], conic[1, 1]
        px, py = rng.uniform(0, 64, 2)
        for tx in range(-1, 9):
            for ty in range(-1, 9):
                xs = np.arange(tx * 8, tx * 8 + 8) - px
                ys = np.arange(ty * 8, ty * 8 + 8) - py
                dx, dy = np.meshgrid(xs, ys)
                q_pix = (A * dx * dx + 2 * B * dx * dy + C * dy * dy).min()
                fx, fy = np.meshgrid(np.linspace(xs[0], xs[-1], 57), np.linspace(ys[0], ys[-1], 57))
                q_box = (A * fx * fx + 2 * B * fx * fy + C * fy * fy).min()
                m = qmin(px, py, A, B, C, tx, ty)
                assert m <= q_pix * (1 + 1e-9) + 1e-12                     # lower bound of every pixel of the bin
                # ... and it IS the box minimum (the 57 x 57 sample grid misses the true minimiser by <= 1/16 px per axis)
                assert m <= q_box * (1 + 1e-9) + 1e-12 and m >= q_box - 0.05 * q_box - (A + C + 2 * abs(B)) / 128.0
