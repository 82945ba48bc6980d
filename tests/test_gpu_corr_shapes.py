"""GPU parity of the correlation volume build and the lookups AT THE MAP SHAPES OF BASELINE.json's configs:
64x64 (TUM-VI 512x512, all 96 edges of the 25-KF window), 28x107 (KITTI-360), 55x55 (TUM-VI at the demo's
resolution) and 48x64 (WHU), all with the 128 feature channels of the real encoder.

Per shape: (1) the volume pyramid of a sample of edges against the CPU oracle and the exact float64 product, and the
flow-aligned ("sheared") pyramid as a bit-exact re-indexing of it; (2) the fused sheared lookup, the fused
reference-layout lookup and the drop-in droid_backends.corr_index_forward, bit-exact against the oracle on the sample
edges and bit-exact against each other on ALL edges."""
import numpy as np
import pytest
import torch

from dbaf_amd import synthetic as syn

pytestmark = pytest.mark.gpu

SHAPES = {
    # id: (h, w, keyframes, band radius, extra pairs, edges checked against the oracle)
    "tumvi_room1_64x64_C128_96edges": (64, 64, 25, 2, [(0, 3)], (0, 47, 95)),
    "kitti360_28x107_C128": (28, 107, 6, 2, [], (0, 9, 17)),
    "tumvi_corridor_55x55_C128": (55, 55, 5, 2, [], (0, 6, 13)),
    "whu_48x64_C128": (48, 64, 5, 2, [], (1, 8, 12)),
}
INTR = {(64, 64): syn.TUMVI_INTRINSICS_8, (28, 107): (69.0, 69.5, 53.2, 14.1), (55, 55): (20.5, 20.5, 27.4, 27.6),
        (48, 64): (30.0, 30.0, 31.5, 23.7)}


def _oracle():
    from oracle import oracle as orc
    return orc


def _setup(name):
    h, w, kf, rad, extra, sample = SHAPES[name]
    ii, jj = syn.graph_banded(kf, rad, extra=extra)
    W = syn.make_window(ii, jj, kf, h, w, seed=5, intr=INTR[(h, w)])
    fmaps = syn.make_fmaps(W.B, 128, h, w, 1005)
    coords = syn.lookup_coords(W, oob_frac=0.05, seed=5)          # [N, h, w, 2] coherent flow + 5 % thrown far out
    coords[0, 0, 0] = (3.0, 4.0)                                   # integer coordinates
    coords[0, 0, 1] = (-0.5, h - 0.5)                              # straddles two borders
    coords[0, 0, 2] = (np.nan, 1.0)
    return W, fmaps, coords, [s for s in sample if s < W.N]


def _ulp_diff_f16(a, b):
    ai = a.view(np.int16).astype(np.int32)
    bi = b.view(np.int16).astype(np.int32)
    ai = np.where(ai < 0, -(ai & 0x7fff), ai)
    bi = np.where(bi < 0, -(bi & 0x7fff), bi)
    return np.abs(ai - bi)


@pytest.mark.parametrize("name", list(SHAPES))
def test_volume_pyramid_at_config_shape(name):
    from dbaf_amd.corr import CorrBlock
    orc = _oracle()
    W, fmaps, _, sample = _setup(name)
    h, w = W.h, W.w
    f = torch.from_numpy(fmaps).cuda()
    sel = torch.as_tensor(sample)
    f1, f2 = f[torch.from_numpy(W.ii)[sel]][None], f[torch.from_numpy(W.jj)[sel]][None]
    ref_pyr = CorrBlock.build_pyramid(f1, f2, 4)
    cb = CorrBlock(f1, f2, num_levels=4, radius=3, layout="sheared")
    for k, e in enumerate(sample):
        a1, a2 = fmaps[W.ii[e]][None], fmaps[W.jj[e]][None]
        want = orc.corr_pyramid(a1, a2, 4)
        g0 = ref_pyr[0][k].cpu().numpy()
        a = (a1[0].astype(np.float64) / 4).reshape(128, h * w)
        b = (a2[0].astype(np.float64) / 4).reshape(128, h * w)
        exact = (a.T @ b).reshape(g0.shape)
        bound = 128 * 6e-8 * (np.abs(a).T @ np.abs(b)).reshape(g0.shape)
        ulp16 = np.maximum(np.spacing(np.abs(exact).astype(np.float16)).astype(np.float64), 2.0 ** -24)
        assert (np.abs(g0.astype(np.float64) - exact) <= 0.5 * ulp16 + bound).all()
        ulp = _ulp_diff_f16(g0, want[0][0])
        assert (ulp > 0).mean() < 0.01 and ulp[np.abs(want[0][0].astype(np.float32)) > 0.01].max() <= 1
        for lvl in range(1, 4):   # pooled levels: exact functions of the (rounded) level below
            below = ref_pyr[lvl - 1][k:k + 1].cpu().numpy()
            got = ref_pyr[lvl][k:k + 1].cpu().numpy()
            assert np.array_equal(got.view(np.uint16), orc.avg_pool2(below).view(np.uint16))
        for lvl in range(4):      # sheared pyramid (fused build where the shape allows) == re-indexed reference pyramid
            v = ref_pyr[lvl][k].cpu().numpy()
            vs = cb.sheared_level(lvl)[k].cpu().numpy()     # [h2l, w2l, h1, w1]
            hl, wl = v.shape[2], v.shape[3]
            x1, ty, tx = np.meshgrid(np.arange(w), np.arange(hl), np.arange(wl), indexing="ij")
            dx = (tx - (x1 >> lvl)) % wl
            for y1 in range(h):
                dy = (ty - (y1 >> lvl)) % hl
                assert np.array_equal(vs[dy, dx, y1, x1].view(np.uint16), v[y1][x1, ty, tx].view(np.uint16)), (lvl, y1)


@pytest.mark.parametrize("name", list(SHAPES))
def test_lookups_at_config_shape(name, lookup_kernel):
    import droid_backends
    from dbaf_amd.corr import CorrBlock
    orc = _oracle()
    W, fmaps, coords, sample = _setup(name)
    h, w, N = W.h, W.w, W.N
    f = torch.from_numpy(fmaps).cuda()
    ii, jj = torch.from_numpy(W.ii).cuda(), torch.from_numpy(W.jj).cuda()
    cs = cr = None
    for c0 in range(0, N, 32):   # chunked like covisible_graph.add_factors does (a few edges at a time), then cat
        s_ = slice(c0, min(c0 + 32, N))
        a = CorrBlock(f[ii[s_]][None], f[jj[s_]][None], num_levels=4, radius=3, layout="sheared")
        b = CorrBlock(f[ii[s_]][None], f[jj[s_]][None], num_levels=4, radius=3, layout="reference")
        cs = a if cs is None else cs.cat(a)
        cr = b if cr is None else cr.cat(b)
    cdev = torch.from_numpy(coords)[None].cuda()
    out_s = cs(cdev)[0]
    out_r = cr(cdev)[0]
    assert out_s.shape == (N, 196, h, w) and out_s.dtype == torch.float16
    # every edge: the two fused lookups and the per-level drop-in agree bit for bit
    assert torch.equal(out_s.view(torch.int16), out_r.view(torch.int16)), float((out_s != out_r).float().mean())
    cperm = cdev[0].permute(0, 3, 1, 2).contiguous()
    for lvl in range(4):
        o, = droid_backends.corr_index_forward(cr.corr_pyramid[lvl], (cperm / 2 ** lvl).contiguous(), 3)
        assert torch.equal(o.view(N, 49, h, w).view(torch.int16), out_r[:, 49 * lvl:49 * lvl + 49].view(torch.int16))
    # sample edges: against the CPU oracle's lookup (on the reference-layout pyramid the GPU built; the volume itself is
    # checked in test_volume_pyramid_at_config_shape)
    got = out_s.cpu().numpy()
    for e in sample:
        pyr = [p[e:e + 1].cpu().numpy() for p in cr.corr_pyramid]
        ref = orc.corr_lookup_pyramid(pyr, coords[e:e + 1], 3)
        assert np.array_equal(got[e:e + 1].view(np.uint16), ref.view(np.uint16)), (name, e, (got[e:e + 1] != ref).mean())
    assert np.isfinite(got.astype(np.float32)).all() and (got[0, :, 0, 2] == 0).all()


@pytest.mark.parametrize("name", list(SHAPES))
def test_zero_edit_route_is_served_from_a_flow_aligned_shadow_bit_for_bit(name):
    """The reference's own CorrBlock.__call__ (modules/corr.py:40-50: one corr_index_forward per level on the reference
    layout, coords / 2^l, cat) against droid_backends with no import swapped: the first lookup of a level runs on the
    reference layout, the second builds a flow-aligned shadow of the level and is served from it, later ones hit it -- all
    bit-identical to the direct kernel; an in-place write or a new tensor drops the shadow."""
    import droid_backends
    from droid_backends import _SHADOWS
    from dbaf_amd.corr import CorrBlock
    W, fmaps, coords, _ = _setup(name)
    f = torch.from_numpy(fmaps).cuda()
    ii, jj = torch.from_numpy(W.ii).cuda(), torch.from_numpy(W.jj).cuda()
    pyr = CorrBlock.build_pyramid(f[ii][None], f[jj][None], 4)          # reference layout [n, h1, w1, h2l, w2l]
    c = torch.from_numpy(coords).cuda().permute(0, 3, 1, 2).contiguous()  # [n, 2, h, w]

    def call():
        return [droid_backends.corr_index_forward(pyr[l], c / 2 ** l, 3)[0] for l in range(4)]

    assert _SHADOWS.enabled
    _SHADOWS.enabled = False
    try:
        direct = call()
    finally:
        _SHADOWS.enabled = True
    b0, h0 = _SHADOWS.builds, _SHADOWS.hits
    uses = _SHADOWS._min_uses
    _SHADOWS.min_uses = 2          # (the default waits for 13 lookups of the same tensors: the ski-rental rule)
    first, second, third = call(), call(), call()
    assert _SHADOWS.builds == b0 + 4 and _SHADOWS.hits == h0 + 4      # built at the second use, hit at the third
    for l in range(4):
        for got in (first[l], second[l], third[l]):
            assert torch.equal(got.view(torch.int16), direct[l].view(torch.int16)), l
    # an in-place write drops the shadow of that level (a stale one would return the old taps)
    pyr[1][0, 0, 0].zero_()
    _SHADOWS.enabled = False
    try:
        direct1 = droid_backends.corr_index_forward(pyr[1], c / 2, 3)[0]
    finally:
        _SHADOWS.enabled = True
    again = [droid_backends.corr_index_forward(pyr[1], c / 2, 3)[0] for _ in range(3)]
    for got in again:
        assert torch.equal(got.view(torch.int16), direct1.view(torch.int16))
    # a dead tensor takes its shadow with it
    n_before = len(_SHADOWS.seen)
    del pyr, first, second, third, again
    import gc
    gc.collect()
    assert len(_SHADOWS.seen) <= n_before - 4
    _SHADOWS._min_uses = uses


def test_padded_tile_grids_are_an_opt_in_that_changes_no_result():
    """DBA_SHEAR_PAD=1 (read once per process, so: a process of its own) tiles the planes of maps that are within 25 % of a grid of
    whole 4 x 64-pixel bands -- 28 x 107 on 28 x 128, 55 x 55 on 56 x 64 -- and serves them with the rows-over-tiles lookup and the
    tiled build walks: every parity test of the config shapes, of the fused build against the unfused pipeline and of the slot-addressed
    block passes there too, bit for bit (the pad pixels' entries are never read as taps and never returned)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DBA_SHEAR_PAD="1")
    probe = ("import sys, ctypes; sys.path.insert(0, %r); from dbaf_amd import _lib; lib = _lib.load(); a = ctypes.c_int(); b = ctypes.c_int();"
             "t = lib.dba_corr_sheared_grid(28, 107, ctypes.byref(a), ctypes.byref(b)); print(t, a.value, b.value, lib.dba_corr_sheared_plane_elems(55, 55))"
             % os.path.join(root, "dba-fusion_amd"))
    out = subprocess.run([sys.executable, "-c", probe], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[-4:] == ["16", "28", "128", str(56 * 64)], out.stdout
    run = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                          os.path.join(root, "tests", "test_gpu_corr_shapes.py"), os.path.join(root, "tests", "test_gpu_corr.py"),
                          os.path.join(root, "tests", "test_gpu_corr_slots.py"),
                          "-k", "config_shape or fused_sheared_build or sheared_volume_is or cat_and_index or looked_up_once"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-1000:]
    assert " passed" in run.stdout
