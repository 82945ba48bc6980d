"""GPU parity for the correlation half: lookup (bit-exact), volume build (<= 1 fp16 ulp), altcorr."""
import numpy as np
import pytest
import torch

from dbaf_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle as orc
    return orc


def _coords(rng, n, h1, w1, h2, w2, oob=0.1):
    c = np.stack([rng.uniform(-2, w2 + 1, size=(n, h1, w1)), rng.uniform(-2, h2 + 1, size=(n, h1, w1))], 1)
    far = rng.uniform(size=(n, h1, w1)) < oob
    c[:, 0][far] += rng.choice([-1, 1], size=int(far.sum())) * rng.uniform(10, 500, size=int(far.sum()))
    c[0, :, 0, 0] = [3.0, 4.0]        # integer coordinates: dx = dy = 0
    c[0, :, 0, 1] = [-0.5, h2 - 0.5]  # straddles the border
    return c.astype(np.float32)


@pytest.mark.parametrize("dtype", [np.float16, np.float32])
@pytest.mark.parametrize("shape", [(2, 16, 16, 16, 16, 3), (3, 7, 13, 9, 21, 3), (1, 8, 8, 8, 8, 3),
                                   (2, 6, 5, 12, 10, 2), (1, 5, 6, 11, 9, 4), (2, 4, 4, 20, 20, 1)])
def test_corr_index_forward_bit_exact(dtype, shape):
    import droid_backends
    orc = _oracle()
    n, h1, w1, h2, w2, r = shape
    rng = np.random.default_rng(hash(shape) % 1000)
    vol = (rng.standard_normal((n, h1, w1, h2, w2)) * 4).astype(dtype)
    coords = _coords(rng, n, h1, w1, h2, w2)
    ref = orc.corr_index_forward(vol, coords, r)
    out, = droid_backends.corr_index_forward(torch.from_numpy(vol).cuda(), torch.from_numpy(coords).cuda(), r)
    got = out.cpu().numpy()
    assert got.shape == ref.shape and got.dtype == ref.dtype
    if dtype == np.float16:
        assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), \
            "max abs diff %g" % np.abs(got.astype(np.float32) - ref.astype(np.float32)).max()
    else:
        assert np.array_equal(got, ref), "max abs diff %g" % np.abs(got - ref).max()


def test_corr_index_forward_empty_and_errors():
    import droid_backends
    vol = torch.zeros(0, 4, 4, 4, 4, dtype=torch.float16, device="cuda")
    coords = torch.zeros(0, 2, 4, 4, device="cuda")
    out, = droid_backends.corr_index_forward(vol, coords, 3)
    assert out.shape == (0, 7, 7, 4, 4)
    with pytest.raises(RuntimeError):
        droid_backends.corr_index_forward(torch.zeros(1, 4, 4, 4, 4), torch.zeros(1, 2, 4, 4), 3)  # CPU tensors


def _smooth_coords(rng, n, h, w):
    """coherent flow (as in real use) plus a few incoherent / out-of-bounds pixels: [n,h,w,2]"""
    y, x = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    c = np.zeros((n, h, w, 2), np.float32)
    for e in range(n):
        fx, fy = rng.uniform(-6, 6, size=2)
        c[e, ..., 0] = x * (1 + rng.uniform(-0.05, 0.05)) + fx + 0.3 * np.sin(0.3 * y + e)
        c[e, ..., 1] = y * (1 + rng.uniform(-0.05, 0.05)) + fy + 0.3 * np.cos(0.2 * x)
    wild = rng.uniform(size=(n, h, w)) < 0.03
    c[wild] += rng.uniform(-300, 300, size=(int(wild.sum()), 2)).astype(np.float32)
    c[0, 0, 1] = [3.0e5, -3.0e5]  # far out of bounds, still int-representable
    return c


@pytest.mark.parametrize("layout", ["reference", "sheared"])
@pytest.mark.parametrize("shape", [(3, 32, 16, 24), (2, 16, 24, 64), (1, 16, 16, 136), (2, 16, 20, 44), (2, 16, 17, 23),
                                   (1, 16, 18, 71), (2, 16, 12, 128), (1, 16, 8, 192), (2, 16, 28, 107), (1, 16, 16, 136)])
def test_corrblock_pyramid_lookup_matches_per_level_oracle(layout, shape, lookup_kernel):
    if layout == "reference" and lookup_kernel != "auto":
        pytest.skip("the kernel selection only concerns the sheared layout")
    """fused CorrBlock.__call__ == 4x corr_index_forward(coords / 2^l) + cat (corr.py:40-50), for both
    internal volume layouts, on random and on coherent coordinates"""
    from dbaf_amd.corr import CorrBlock
    orc = _oracle()
    rng = np.random.default_rng(5)
    n, C, h, w = shape
    f1 = rng.standard_normal((1, n, C, h, w)).astype(np.float16)
    f2 = rng.standard_normal((1, n, C, h, w)).astype(np.float16)
    t1, t2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
    cb = CorrBlock(t1, t2, num_levels=4, radius=3, layout=layout)
    pyr_ref = [p.cpu().numpy() for p in CorrBlock.build_pyramid(t1, t2, 4)]
    for coords in (_coords(rng, n, h, w, h, w).transpose(0, 2, 3, 1), _smooth_coords(rng, n, h, w)):
        coords = np.ascontiguousarray(coords)
        out = cb(torch.from_numpy(coords)[None].cuda()).cpu().numpy()[0]
        ref = orc.corr_lookup_pyramid(pyr_ref, coords, 3)  # oracle lookup on the GPU-built reference pyramid
        assert out.shape == ref.shape == (n, 196, h, w)
        assert np.array_equal(out.view(np.uint16), ref.view(np.uint16)), (layout, (out != ref).mean())


@pytest.mark.parametrize("layout", ["reference", "sheared"])
def test_lookup_non_finite_coords_do_not_fault(layout, lookup_kernel):
    if layout == "reference" and lookup_kernel != "auto":
        pytest.skip("the kernel selection only concerns the sheared layout")
    """NaN / inf / 1e30 coordinates (undefined behaviour in the reference's float->int cast) read nothing:
    the HIP path treats them as entirely out of bounds and must neither fault nor disturb other pixels."""
    from dbaf_amd.corr import CorrBlock
    rng = np.random.default_rng(8)
    n, C, h, w = 1, 16, 16, 64
    t1 = torch.from_numpy(rng.standard_normal((1, n, C, h, w)).astype(np.float16)).cuda()
    t2 = torch.from_numpy(rng.standard_normal((1, n, C, h, w)).astype(np.float16)).cuda()
    cb = CorrBlock(t1, t2, num_levels=4, radius=3, layout=layout)
    coords = _smooth_coords(rng, n, h, w)
    good = cb(torch.from_numpy(coords)[None].cuda()).cpu().numpy()[0]
    bad = coords.copy()
    bad[0, 3, 5] = [np.nan, 2.0]
    bad[0, 3, 6] = [np.inf, -np.inf]
    bad[0, 3, 7] = [1e30, -1e30]
    out = cb(torch.from_numpy(bad)[None].cuda()).cpu().numpy()[0]
    assert np.isfinite(out.astype(np.float32)).all()
    assert (out[0, :, 3, 5:8] == 0).all()
    mask = np.ones((h, w), bool)
    mask[3, 5:8] = False
    assert np.array_equal(out[0][:, mask].view(np.uint16), good[0][:, mask].view(np.uint16))


@pytest.mark.parametrize("shape", [(2, 128, 64, 64), (3, 32, 24, 64), (1, 16, 8, 64), (2, 128, 28, 107), (1, 32, 55, 55),
                                   (1, 16, 48, 64), (1, 16, 18, 71), (2, 16, 20, 44), (1, 16, 9, 128), (1, 16, 16, 16), (1, 16, 12, 128),
                                   (1, 16, 24, 107), (2, 128, 28, 107), (1, 32, 8, 120),
                                   (2, 16, 8, 8), (1, 16, 9, 10), (1, 32, 11, 13), (1, 16, 8, 65), (1, 16, 33, 36),
                                   (1, 128, 55, 55), (2, 128, 18, 44), (3, 128, 9, 10), (1, 128, 11, 13), (5, 128, 8, 8),
                                   (2, 128, 24, 40), (1, 128, 20, 32), (3, 128, 8, 16), (1, 128, 30, 56), (2, 128, 12, 48)])
def test_fused_sheared_build_equals_unfused_pipeline(shape):
    """one-pass MFMA build (GEMM + pooling + shear) == GEMM kernel + 3 pooling passes + shear passes, bit for bit, for
    64-wide maps, maps whose strips span row ends (107, 55, 71, 44 wide), heights that are not multiples of 8 (28,
    55, 18, 20, 9: partial target tiles and the floor sizes of avg_pool2d), the widest supported map (128), the
    smallest ones (8 x 8, 9 x 10, 11 x 13: the store loops' column counters wrap more than once per line there),
    widths just past a tile size (65, 36), C = 128 on irregular maps (the strip-walking form with row-end quads), and C = 128
    on maps whose width is a multiple of 8 (operands read straight from the [n, C, h, w] maps: linear and tiled pixel order)"""
    from dbaf_amd.corr import CorrBlock
    n, C, h, w = shape
    rng = np.random.default_rng(12)
    t1 = torch.from_numpy(rng.standard_normal((1, n, C, h, w)).astype(np.float16)).cuda()
    t2 = torch.from_numpy(rng.standard_normal((1, n, C, h, w)).astype(np.float16)).cuda()
    fused = CorrBlock.build_sheared_fused(t1, t2, 4)
    assert fused is not None
    unfused = CorrBlock.shear_pyramid(CorrBlock.build_pyramid(t1, t2, 4))
    for lvl in range(4):
        assert fused[lvl].shape == unfused[lvl].shape and fused[lvl].shape[-1] % 64 == 0   # [n, h2l, w2l, HW1p]
        # (the plane padding is never written; tiled planes -- also with tiles that reach past the map -- come back row-major)
        a = CorrBlock.map_pixels(fused[lvl], h, w).cpu().numpy()
        b = CorrBlock.map_pixels(unfused[lvl], h, w).cpu().numpy()
        assert np.array_equal(a.view(np.uint16), b.view(np.uint16)), (lvl, (a != b).mean())


@pytest.mark.parametrize("n,h,w", [(1, 64, 64), (5, 64, 64), (3, 8, 64), (2, 40, 64), (7, 24, 64), (33, 16, 64), (2, 48, 64),
                                   (1, 55, 55), (5, 55, 55), (3, 9, 33), (2, 44, 60), (33, 17, 37), (7, 8, 63), (2, 61, 61),
                                   (5, 28, 107), (1, 28, 107), (2, 9, 126), (3, 17, 125), (1, 12, 127), (2, 20, 65)])
def test_sixteen_wave_build_every_walk_length_and_into_slots(n, h, w):
    """the sixteen-wave strip walks (C = 128; 64-wide maps whose planes are tiled: corr_build_fused16_kernel; maps 33..63 wide whose
    planes keep the linear pixel order and whose 16-byte pieces are not aligned, 55 x 55 of the reference's TUM-VI demo among them:
    corr_build_fused16g_kernel; maps 65..128 wide in the linear order, 28 x 107 of the KITTI config among them: corr_build_fused16w_kernel,
    one strip per workgroup, incl. the widths from 125 on whose wrap columns lie outside the second wave's half) at the walk lengths
    the launch heuristic produces -- one edge (two strips per workgroup), walks that
    end early (the last workgroup of a row tile holds fewer strips), one row tile (h = 8), a last row tile with one valid row
    (h = 9, 17), widths just past 32 and just below 64 (the split of the level-0 lines between a row's two waves), more edges than
    one round of workgroups -- bit for bit against the unfused pipeline; then the same edges built INTO THE SLOTS of a standing
    block (holes in arbitrary order): every slot's planes equal the fresh build's, untouched slots keep their bytes"""
    from dbaf_amd.corr import CorrBlock
    C = 128
    rng = np.random.default_rng(100 + n + h + (w - 64))
    t1 = torch.from_numpy(rng.standard_normal((1, n, C, h, w)).astype(np.float16)).cuda()
    t2 = torch.from_numpy(rng.standard_normal((1, n, C, h, w)).astype(np.float16)).cuda()
    fused = CorrBlock.build_sheared_fused(t1, t2, 4)
    unfused = CorrBlock.shear_pyramid(CorrBlock.build_pyramid(t1, t2, 4))
    for lvl in range(4):
        a = CorrBlock.map_pixels(fused[lvl], h, w).contiguous().view(torch.int16)
        b = CorrBlock.map_pixels(unfused[lvl], h, w).contiguous().view(torch.int16)
        assert torch.equal(a, b), lvl
    # a standing block of n + 3 old edges; drop n of them (every second one first), cat the new ones into the holes
    m = n + 3
    o1 = torch.from_numpy(rng.standard_normal((1, m, C, h, w)).astype(np.float16)).cuda()
    o2 = torch.from_numpy(rng.standard_normal((1, m, C, h, w)).astype(np.float16)).cuda()
    blk = CorrBlock(o1, o2, capacity=m + 1).build()
    old = [s.clone() for s in blk._stores]
    drop = np.zeros(m, bool)
    drop[rng.permutation(m)[:n]] = True
    keep = torch.from_numpy(~drop).cuda()
    blk = blk[keep].cat(CorrBlock(t1, t2))
    assert blk.stats["copied_edges"] == 0 and blk.stats["grown"] == 0
    slots = blk._slots.cpu().numpy()
    assert len(set(slots.tolist())) == 3 + n
    kept_old = np.flatnonzero(~drop)
    for lvl in range(4):
        st = blk._stores[lvl].view(torch.int16)
        for k in range(3):                                   # the surviving old edges: same slot, same bytes
            assert slots[k] == kept_old[k]
            assert torch.equal(st[slots[k]], old[lvl].view(torch.int16)[kept_old[k]])
        for e in range(n):                                   # the new edges, wherever they landed
            assert torch.equal(CorrBlock.map_pixels(st[slots[3 + e]][None], h, w), CorrBlock.map_pixels(fused[lvl][e][None].view(torch.int16), h, w)), (lvl, e)
        free = sorted(set(range(m + 1)) - set(slots.tolist()))
        assert free == [m]                                    # the spare slot was never written
        assert torch.equal(st[m], old[lvl].view(torch.int16)[m])


def test_sheared_volume_is_a_permutation_of_the_reference_volume():
    from dbaf_amd.corr import CorrBlock
    rng = np.random.default_rng(6)
    n, C, h, w = 2, 16, 16, 24
    t1 = torch.from_numpy(rng.standard_normal((1, n, C, h, w)).astype(np.float16)).cuda()
    t2 = torch.from_numpy(rng.standard_normal((1, n, C, h, w)).astype(np.float16)).cuda()
    ref = CorrBlock.build_pyramid(t1, t2, 3)
    shr = CorrBlock.shear_pyramid(ref)
    for lvl, (v, vs) in enumerate(zip(ref, shr)):
        v, vs = v.cpu().numpy(), vs.cpu().numpy()
        assert vs.shape[-1] == 384 and vs.shape[-1] % 64 == 0   # 16 x 24 pixels: already a multiple of 64
        vs = vs.reshape(vs.shape[:3] + (h, w))
        hl, wl = v.shape[3], v.shape[4]
        y1, x1, ty, tx = np.meshgrid(np.arange(h), np.arange(w), np.arange(hl), np.arange(wl), indexing="ij")
        dy, dx = (ty - (y1 >> lvl)) % hl, (tx - (x1 >> lvl)) % wl
        for e in range(n):
            assert np.array_equal(vs[e][dy, dx, y1, x1], v[e][y1, x1, ty, tx])


def _ulp_diff_f16(a, b):
    ai = a.view(np.int16).astype(np.int32)
    bi = b.view(np.int16).astype(np.int32)
    ai = np.where(ai < 0, -(ai & 0x7fff), ai)
    bi = np.where(bi < 0, -(bi & 0x7fff), bi)
    return np.abs(ai - bi)


@pytest.mark.parametrize("shape", [(2, 128, 16, 16), (1, 64, 11, 13), (1, 128, 8, 40)])
def test_corr_volume_build_matches_oracle(shape):
    from dbaf_amd.corr import CorrBlock
    orc = _oracle()
    n, C, h, w = shape
    rng = np.random.default_rng(7)
    f1 = rng.standard_normal((n, C, h, w)).astype(np.float16)
    f2 = rng.standard_normal((n, C, h, w)).astype(np.float16)
    nl = 4 if min(h, w) >= 16 else 2
    pyr = CorrBlock.build_pyramid(torch.from_numpy(f1)[None].cuda(), torch.from_numpy(f2)[None].cuda(), nl)
    ref = orc.corr_pyramid(f1, f2, nl)
    # level 0 = fp16(round) of an fp32-accumulated 128-term dot product.  The MFMA accumulation order differs
    # from any sequential order, so compare with the exact (float64) product: half an fp16 ulp of rounding
    # plus the fp32 accumulation error bound K * eps32 * sum|a b| (here ~1e-5).
    g0 = pyr[0].cpu().numpy()
    assert g0.shape == ref[0].shape
    a = (f1.astype(np.float64) / 4).reshape(n, C, h * w)
    b = (f2.astype(np.float64) / 4).reshape(n, C, h * w)
    exact = np.einsum("ncp,ncq->npq", a, b).reshape(g0.shape)
    bound = C * 6e-8 * np.einsum("ncp,ncq->npq", np.abs(a), np.abs(b)).reshape(g0.shape)
    ulp16 = np.maximum(np.spacing(np.abs(exact).astype(np.float16)).astype(np.float64), 2.0 ** -24)
    assert (np.abs(g0.astype(np.float64) - exact) <= 0.5 * ulp16 + bound).all()
    ulp = _ulp_diff_f16(g0, ref[0])  # and against the sequential-order oracle: identical but for rare 1-ulp ties
    assert (ulp > 0).mean() < 0.01 and ulp[np.abs(ref[0].astype(np.float32)) > 0.01].max() <= 1
    # pooled levels are exact functions of the level below: check against the oracle pooling of the GPU level
    for l in range(1, nl):
        below = pyr[l - 1].cpu().numpy()
        want = orc.avg_pool2(below)
        got = pyr[l].cpu().numpy()
        assert got.shape == want.shape
        assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


def _alt_case(rng, B, S, H1, W1, H2, W2, C, coherent):
    f1 = rng.standard_normal((B, H1, W1, C)).astype(np.float32)
    f2 = rng.standard_normal((B, H2, W2, C)).astype(np.float32)
    if coherent:   # smooth flow: the tile's windows overlap, the LDS-staged path runs
        yy, xx = np.meshgrid(np.arange(H1, dtype=np.float32), np.arange(W1, dtype=np.float32), indexing="ij")
        coords = np.zeros((B, S, H1, W1, 2), np.float32)
        for b in range(B):
            for s in range(S):
                coords[b, s, ..., 0] = xx * (W2 / W1) + rng.uniform(-3, 3) + 0.4 * np.sin(0.3 * yy + s)
                coords[b, s, ..., 1] = yy * (H2 / H1) + rng.uniform(-3, 3) + 0.4 * np.cos(0.2 * xx + b)
        wild = rng.uniform(size=(B, S, H1, W1)) < 0.02
        coords[wild] += rng.uniform(-200, 200, size=(int(wild.sum()), 2)).astype(np.float32)
    else:          # incoherent: every lane reads its own taps
        coords = np.stack([rng.uniform(-2, W2 + 1, size=(B, S, H1, W1)), rng.uniform(-2, H2 + 1, size=(B, S, H1, W1))],
                          -1).astype(np.float32)
    coords[0, 0, 0, 0] = (2.0, 1.0)              # integer coordinates
    coords[0, 0, 0, 1] = (-0.5, H2 - 0.5)        # straddles two borders
    return f1, f2, coords


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("case", [
    # B, S, H1, W1, H2, W2, C, radius, coherent
    (2, 2, 9, 12, 5, 7, 64, 3, False),
    (2, 1, 24, 40, 24, 40, 128, 3, True),      # the AltCorrBlock call shape: level 0, 128 channels
    (1, 2, 24, 40, 12, 20, 128, 3, True),      # level 1 (pooled fmap2)
    (1, 1, 13, 21, 13, 21, 32, 1, True),
    (1, 1, 13, 21, 13, 21, 32, 2, True),
    (1, 1, 10, 18, 10, 18, 32, 4, True),
    (1, 1, 7, 9, 6, 5, 96, 2, False),
    (1, 1, 8, 16, 8, 16, 36, 3, True),         # channel count that is not a multiple of a slice
], ids=lambda c: "B%dS%d_%dx%d_from_%dx%d_C%d_r%d_%s" % (c[:8] + ("coh" if c[8] else "inc",)))
def test_altcorr_forward_matches_oracle(case, dtype):
    """float: the oracle accumulates with separate multiply and add, the kernel with the fused multiply-add nvcc emits for
    the reference's `s += a * b` -> a few float32 ulps; half: c10::Half arithmetic on both sides -> bit-exact"""
    import droid_backends
    orc = _oracle()
    B, S, H1, W1, H2, W2, C, r, coherent = case
    rng = np.random.default_rng(9)
    f1, f2, coords = _alt_case(rng, B, S, H1, W1, H2, W2, C, coherent)
    f1, f2 = f1.astype(dtype), f2.astype(dtype)
    ref = orc.altcorr_forward(f1, f2, coords, r)
    out, = droid_backends.altcorr_forward(torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda(),
                                          torch.from_numpy(coords).cuda(), r)
    got = out.cpu().numpy()
    assert got.shape == ref.shape and got.dtype == ref.dtype
    if dtype == np.float16:
        assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), \
            (float((got != ref).mean()), float(np.abs(got.astype(np.float32) - ref.astype(np.float32)).max()))
    else:
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5 * np.sqrt(C / 64))


def test_altcorrblock_mirror_matches_per_level_oracle():
    """dbaf_amd.corr.AltCorrBlock (mirror of modules/corr.py:91-139) == per-level altcorr on the pooled channels-last
    pyramid, concatenated; and it agrees with the volume-based CorrBlock lookup to rounding (same quantity, no volume)"""
    from dbaf_amd.corr import AltCorrBlock, CorrBlock
    orc = _oracle()
    rng = np.random.default_rng(19)
    B, N, C, H, W = 1, 3, 128, 24, 32
    fmaps = (0.5 * rng.standard_normal((B, N, C, H, W))).astype(np.float32)
    ii, jj = np.array([0, 1, 2, 0]), np.array([1, 2, 0, 2])
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    coords = np.stack([xx[None] + rng.uniform(-2, 2, size=(4, 1, 1)), yy[None] + rng.uniform(-2, 2, size=(4, 1, 1))],
                      -1).astype(np.float32)[None]                                    # [1, 4, H, W, 2]
    t = torch.from_numpy(fmaps).cuda()
    blk = AltCorrBlock(t, num_levels=4, radius=3)
    out = blk(torch.from_numpy(coords).cuda(), torch.from_numpy(ii).cuda(), torch.from_numpy(jj).cuda())
    assert out.shape == (1, 4, 196, H, W)
    # that was ONE launch for the four levels (no autograd involved); with a gradient wanted the levels go through
    # CorrLayer one by one -- the same arithmetic: identical bits, and the gradient reaches the coordinates' producer
    cg = torch.from_numpy(coords).cuda().requires_grad_(True)
    out_g = blk(cg, torch.from_numpy(ii).cuda(), torch.from_numpy(jj).cuda())
    assert out_g.requires_grad and torch.equal(out_g.detach(), out)
    # two coordinate sets per pixel ([B, N, H, W, S, 2]): the fused launch answers both
    c2 = torch.stack([torch.from_numpy(coords).cuda(), torch.from_numpy(coords).cuda() + 0.37], 4)
    out2 = blk(c2, torch.from_numpy(ii).cuda(), torch.from_numpy(jj).cuda())
    assert out2.shape == (1, 4, 196, H, W, 2) and torch.equal(out2[..., 0], out)
    got = out.cpu().numpy()[0]
    pyr = [p.cpu().numpy()[0] for p in blk.pyramid]                                   # [N, H>>l, W>>l, C] pre-divided by 4
    for lvl in range(4):
        ref = orc.altcorr_forward(pyr[0][ii], pyr[lvl][jj], (coords[0] / 2 ** lvl)[:, None], 3)[:, 0]   # [4, 49, H, W]
        np.testing.assert_allclose(got[:, 49 * lvl:49 * lvl + 49], ref, rtol=2e-5, atol=2e-5)
    # against the materialised volume (half GEMM + half lookup): same quantity to half precision
    cb = CorrBlock(t[:, ii].half(), t[:, jj].half(), num_levels=4, radius=3)
    vol = cb(torch.from_numpy(coords).cuda()).float().cpu().numpy()[0]
    np.testing.assert_allclose(got, vol, rtol=0, atol=0.02 * np.abs(vol).max())


@pytest.mark.parametrize("case", ["smooth", "random", "edges", "wide"])
def test_altcorrblock_on_half_maps_runs_on_the_matrix_cores_and_matches_the_float_chain(case):
    """The reference's case: the feature maps are HALF (autocast), AltCorrBlock keeps fmaps / 4 and its 2x2 averages in half
    and casts them with .float() at every lookup (modules/corr.py:98-120).  dbaf_amd.corr.AltCorrBlock then feeds the halves
    to the matrix cores (dba_altcorr_pyramid_forward_f16maps: exact products, float sums): equal to the float chain on the
    .float() maps up to the order of the float additions, on coherent flow (one small GEMM per 4 x 16 tile), on incoherent
    coordinates (per-thread dot products), at the map borders, for NaN / huge coordinates and on maps that are not whole tiles."""
    from dbaf_amd.corr import AltCorrBlock
    orc = _oracle()
    rng = np.random.default_rng(23)
    B, N, C, H, W = (1, 3, 128, 24, 32) if case != "wide" else (1, 2, 64, 18, 71)
    fmaps = (0.5 * rng.standard_normal((B, N, C, H, W))).astype(np.float16)
    ii, jj = np.array([0, 1, 2 % N, 0]), np.array([1, 0, 0, 1])
    E = len(ii)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    if case == "random":
        coords = np.stack([rng.uniform(-6, W + 6, (E, H, W)), rng.uniform(-6, H + 6, (E, H, W))], -1).astype(np.float32)
    else:
        amp = 2.0 if case != "edges" else 14.0
        coords = np.stack([xx[None] + rng.uniform(-amp, amp, (E, 1, 1)) + 0.3 * np.sin(yy / 3)[None],
                           yy[None] + rng.uniform(-amp, amp, (E, 1, 1)) + 0.3 * np.cos(xx / 4)[None]], -1).astype(np.float32)
    coords[0, 0, 0] = (np.nan, 1.0)
    coords[0, 0, 1] = (3.0e9, 2.0)
    coords[0, 0, 2] = (5.0, 7.0)                     # integer coordinates
    coords[1, 3, 4] = (-300.0, 9.0)                  # a window that misses the map must not stretch the tile's box
    t = torch.from_numpy(fmaps).cuda()
    blk = AltCorrBlock(t, num_levels=4, radius=3)
    assert blk.pyramid[0].dtype == torch.float16 and blk.mfma
    cdev = torch.from_numpy(coords)[None].cuda()
    iid, jjd = torch.from_numpy(ii).cuda(), torch.from_numpy(jj).cuda()
    out = blk(cdev, iid, jjd)
    assert out.shape == (1, E, 196, H, W) and out.dtype == torch.float32 and blk._f32 is None   # no float twins were made
    got = out.cpu().numpy()[0]
    pyr = [p.float().cpu().numpy()[0] for p in blk.pyramid]
    for lvl in range(4):
        ref = orc.altcorr_forward(pyr[0][ii], pyr[lvl][jj], (coords / 2 ** lvl)[:, None], 3)[:, 0]
        np.testing.assert_allclose(got[:, 49 * lvl:49 * lvl + 49], ref, rtol=2e-5, atol=2e-5)
    assert np.isnan(got[0, :, 0, 0]).all() and (got[0, :, 0, 1] == 0).all()   # NaN weights / a window far off the map
    # the float route of the same block (DBA_ALTCORR_MFMA=0 / float maps) agrees to the same bound
    blk.mfma = False
    out_f = blk(cdev, iid, jjd).cpu().numpy()[0]
    np.testing.assert_allclose(got, out_f, rtol=2e-5, atol=2e-5)
    # two coordinate sets per pixel
    blk.mfma = True
    c2 = torch.stack([cdev, cdev + 0.37], 4)
    out2 = blk(c2, iid, jjd)
    assert out2.shape == (1, E, 196, H, W, 2)
    assert torch.equal(torch.nan_to_num(out2[..., 0], nan=7.0), torch.nan_to_num(out, nan=7.0))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_corr_index_backward_matches_oracle(dtype):
    import droid_backends
    orc = _oracle()
    rng = np.random.default_rng(10)
    n, h1, w1, h2, w2, r = 2, 6, 7, 9, 8, 3
    coords = _coords(rng, n, h1, w1, h2, w2)
    cg = rng.standard_normal((n, 7, 7, h1, w1)).astype(np.float32)
    ref = orc.corr_index_backward((n, h1, w1, h2, w2), coords, cg, r)
    if dtype == torch.float16:
        cg = cg.astype(np.float16).astype(np.float32)
        ref = orc.corr_index_backward((n, h1, w1, h2, w2), coords, cg, r)
    vol = torch.zeros(n, h1, w1, h2, w2, device="cuda", dtype=dtype)   # half volume: the gradient comes back as half
    out, = droid_backends.corr_index_backward(vol, torch.from_numpy(coords).cuda(), torch.from_numpy(cg).cuda().to(dtype), r)
    assert out.dtype == dtype
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_altcorr_backward_matches_oracle(dtype):
    """half maps (the reference dispatches them too, altcorr_kernel.cu:337): the adjoint runs in float and is cast back"""
    import droid_backends
    orc = _oracle()
    rng = np.random.default_rng(11)
    B, S, H1, W1, H2, W2, C, r = 2, 2, 6, 7, 5, 6, 48, 3
    f1 = rng.standard_normal((B, H1, W1, C)).astype(np.float32)
    f2 = rng.standard_normal((B, H2, W2, C)).astype(np.float32)
    coords = np.stack([rng.uniform(-2, W2 + 1, size=(B, S, H1, W1)), rng.uniform(-2, H2 + 1, size=(B, S, H1, W1))],
                      -1).astype(np.float32)
    cg = rng.standard_normal((B, S, 49, H1, W1)).astype(np.float32)
    if dtype == torch.float16:
        f1, f2, cg = (a.astype(np.float16).astype(np.float32) for a in (f1, f2, cg))
    r1, r2 = orc.altcorr_backward(f1, f2, coords, cg, r)
    g1, g2, gc = droid_backends.altcorr_backward(torch.from_numpy(f1).cuda().to(dtype), torch.from_numpy(f2).cuda().to(dtype),
                                                 torch.from_numpy(coords).cuda(), torch.from_numpy(cg).cuda().to(dtype), r)
    assert g1.dtype == dtype and g2.dtype == dtype
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-3, atol=2e-2)
    np.testing.assert_allclose(g1.float().cpu().numpy(), r1, **tol)
    np.testing.assert_allclose(g2.float().cpu().numpy(), r2, **tol)
    assert float(gc.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("n,h", [(1, 8), (1, 24), (3, 8), (5, 24), (7, 40), (11, 8)])
def test_sheared_lookup_with_workgroup_counts_that_are_not_multiples_of_8(n, h, lookup_kernel):
    """the lookup re-maps workgroups to XCDs (csrc/corr_sheared.hip): every row must still be covered exactly once
    for any grid size"""
    from dbaf_amd.corr import CorrBlock
    orc = _oracle()
    rng = np.random.default_rng(n * 100 + h)
    C, w = 16, 64
    f1 = rng.standard_normal((1, n, C, h, w)).astype(np.float16)
    f2 = rng.standard_normal((1, n, C, h, w)).astype(np.float16)
    t1, t2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
    cb = CorrBlock(t1, t2, num_levels=2, radius=3, layout="sheared")
    pyr_ref = [p.cpu().numpy() for p in CorrBlock.build_pyramid(t1, t2, 2)]
    coords = np.ascontiguousarray(_smooth_coords(rng, n, h, w))
    out = cb(torch.from_numpy(coords)[None].cuda()).cpu().numpy()[0]
    ref = orc.corr_lookup_pyramid(pyr_ref, coords, 3)
    assert np.array_equal(out.view(np.uint16), ref.view(np.uint16))


@pytest.mark.parametrize("n", [1, 2, 3, 5, 11, 32])
def test_strip_walking_build_for_every_chunking(n):
    """the strip-walking form of the fused build (64-wide maps, C = 128) hands a workgroup 1, 2, 3, ... 16 consecutive
    strips depending on the number of edges (incl. a last chunk that is shorter): bit for bit the unfused pipeline"""
    from dbaf_amd.corr import CorrBlock
    rng = np.random.default_rng(40 + n)
    h, w, C = 64, 64, 128
    t1 = torch.from_numpy(rng.standard_normal((1, n, C, h, w)).astype(np.float16)).cuda()
    t2 = torch.from_numpy(rng.standard_normal((1, n, C, h, w)).astype(np.float16)).cuda()
    fused = CorrBlock.build_sheared_fused(t1, t2, 4)
    unfused = CorrBlock.shear_pyramid(CorrBlock.build_pyramid(t1, t2, 4))
    for lvl in range(4):
        assert torch.equal(CorrBlock.map_pixels(fused[lvl], h, w).contiguous().view(torch.int16),
                           CorrBlock.map_pixels(unfused[lvl], h, w).contiguous().view(torch.int16)), lvl


def test_strip_walking_build_on_a_48x64_map_with_few_edges():
    from dbaf_amd.corr import CorrBlock
    rng = np.random.default_rng(77)
    h, w, C = 48, 64, 128
    for n in (1, 4):
        t1 = torch.from_numpy(rng.standard_normal((1, n, C, h, w)).astype(np.float16)).cuda()
        t2 = torch.from_numpy(rng.standard_normal((1, n, C, h, w)).astype(np.float16)).cuda()
        fused = CorrBlock.build_sheared_fused(t1, t2, 4)
        unfused = CorrBlock.shear_pyramid(CorrBlock.build_pyramid(t1, t2, 4))
        for lvl in range(4):
            assert torch.equal(CorrBlock.map_pixels(fused[lvl], h, w).contiguous().view(torch.int16),
                               CorrBlock.map_pixels(unfused[lvl], h, w).contiguous().view(torch.int16)), (n, lvl)


@pytest.mark.parametrize("shape", [(1, 64, 64), (2, 64, 64), (1, 55, 55), (1, 28, 107), (2, 24, 40)])
def test_block_built_looked_up_once_and_dropped_takes_one_library_call(shape):
    """MotionFilter.track's pattern (dbaf/motion_filter.py:74-76: `CorrBlock(fmap_kf, fmap_new)(coords0)`, one block per incoming
    frame): the first lookup of a small block that was never built goes through dba_corr_build_lookup_once_sheared (pyramid into
    the stream's standing buffer + lookup, one call; the block stays unbuilt) -- bit-identical to the built block's lookup, frame
    after frame through the same buffer; a second lookup, a `cat` or an index of the same block build it the regular way"""
    from dbaf_amd import corr as corr_mod
    from dbaf_amd.corr import CorrBlock
    n, h, w = shape
    rng = np.random.default_rng(31 + n + h)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    grid = np.stack([xs, ys], -1)[None, None]
    for frame in range(3):
        f1 = torch.from_numpy(rng.standard_normal((1, n, 128, h, w)).astype(np.float16)).cuda()
        f2 = torch.from_numpy(rng.standard_normal((1, n, 128, h, w)).astype(np.float16)).cuda()
        coords = torch.from_numpy((grid + rng.uniform(-2.5, 2.5, (1, n, h, w, 2))).astype(np.float32)).cuda()
        blk = CorrBlock(f1, f2)
        once = blk(coords)
        assert blk._pending is not None and blk._once_used and blk._stores is None          # still unbuilt
        ref = CorrBlock(f1, f2).build()(coords)
        assert torch.equal(once.view(torch.int16), ref.view(torch.int16)), frame
        again = blk(coords)                                                                   # the regular path now
        assert blk._pending is None and blk._stores is not None
        assert torch.equal(again.view(torch.int16), ref.view(torch.int16))
    # a block that served one lookup can still be absorbed by a cat / indexed
    blk = CorrBlock(f1, f2)
    once = blk(coords)
    both = CorrBlock(f1, f2).cat(blk)
    out2 = both(torch.cat([coords, coords], 1))
    assert torch.equal(out2[:, n:].view(torch.int16), once.view(torch.int16))
    # switched off: the regular path from the first lookup on
    old = corr_mod.ONCE_MAX_EDGES
    corr_mod.ONCE_MAX_EDGES = 0
    try:
        blk = CorrBlock(f1, f2)
        out3 = blk(coords)
        assert blk._pending is None and not blk._once_used
        assert torch.equal(out3.view(torch.int16), once.view(torch.int16))
    finally:
        corr_mod.ONCE_MAX_EDGES = old
