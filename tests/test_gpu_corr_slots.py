"""GPU: the slot-addressed CorrBlock (dbaf_amd/corr.py) against the semantics of the reference class it replaces
(/root/reference/dbaf/modules/corr.py:23-60: torch.cat of the pyramid in `cat`, boolean / integer indexing in
`__getitem__`) -- same lookups to the bit, with no volume moved on a graph change -- and the lookup that takes the
reprojection into its launch (dba_corr_lookup_reproject_sheared) against reprojection + lookup."""
import numpy as np
import pytest
import torch

from dbaf_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _fmaps(nf, C, h, w, seed):
    return torch.from_numpy(syn.make_fmaps(nf, C, h, w, seed)).cuda()


def _coords(n, h, w, seed, spread=3.0):
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    c = np.stack([xx, yy], -1)[None] + rng.uniform(-spread, spread, size=(n, 1, 1, 2)) + rng.uniform(-1.5, 1.5, size=(n, h, w, 2))
    c[rng.uniform(size=(n, h, w)) < 0.03] += 500.0          # a few pixels thrown out of the map
    return torch.from_numpy(c.astype(np.float32)).cuda()[None]


@pytest.mark.parametrize("layout,h,w", [("sheared", 24, 64), ("sheared", 20, 28), ("sheared", 8, 128), ("reference", 16, 24)])
def test_cat_and_index_edit_the_slot_table_and_lookups_stay_bit_identical(layout, h, w, lookup_kernel):
    from dbaf_amd.corr import CorrBlock
    C, nf = 32, 7
    fm = _fmaps(nf, C, h, w, 5)
    ii = torch.tensor([0, 1, 1, 2, 2, 3, 3, 4, 5, 6, 6, 0], device="cuda")
    jj = torch.tensor([1, 0, 2, 1, 3, 2, 4, 3, 6, 5, 4, 2], device="cuda")

    def fresh(sel):   # the reference's result: one block over exactly these edges, in this order
        return CorrBlock(fm[ii[sel]][None], fm[jj[sel]][None], layout=layout)

    a = CorrBlock(fm[ii[:5]][None], fm[jj[:5]][None], layout=layout, capacity=10)
    assert a._stores is None                                           # nothing is built before the first use
    b = CorrBlock(fm[ii[5:9]][None], fm[jj[5:9]][None], layout=layout)
    a = a.cat(b)                                                       # add_factors: covisible_graph.py:131
    assert b._stores is None and b._pending is None                    # b was built straight into a's free slots
    assert a.n == 9 and a.capacity == 10 and a.stats == dict(built_edges=9, copied_edges=0, grown=0)
    sel = torch.arange(9, device="cuda")
    c9 = _coords(9, h, w, 1)
    assert torch.equal(a(c9), fresh(sel)(c9))

    mask = torch.tensor([1, 0, 1, 1, 0, 1, 1, 0, 1], dtype=torch.bool, device="cuda")      # rm_factors: :166
    ptrs = [s.data_ptr() for s in a._stores]
    a = a[mask]
    sel = sel[mask]
    assert a.n == 6 and [s.data_ptr() for s in a._stores] == ptrs     # the table was edited, the stores were not touched
    c6 = _coords(6, h, w, 2)
    assert torch.equal(a(c6), fresh(sel)(c6))

    a = a.cat(CorrBlock(fm[ii[9:12]][None], fm[jj[9:12]][None], layout=layout))    # lands in the three freed slots
    sel = torch.cat([sel, torch.arange(9, 12, device="cuda")])
    assert a.n == 9 and a.capacity == 10 and a.stats["grown"] == 0 and a.stats["copied_edges"] == 0
    assert [s.data_ptr() for s in a._stores] == ptrs
    c9b = _coords(9, h, w, 3)
    assert torch.equal(a(c9b), fresh(sel)(c9b))

    perm = torch.tensor([8, 0, 3, 3, 5], device="cuda")               # integer index: re-order, repeat
    a = a[perm]
    sel = sel[perm]
    c5 = _coords(5, h, w, 4)
    assert torch.equal(a(c5), fresh(sel)(c5))
    # the pyramid as the reference exposes it (gathered), level by level
    want = fresh(sel)
    for lvl in range(4):
        if layout == "sheared":    # (without the plane padding, which nobody writes)
            assert torch.equal(a.sheared_level(lvl), want.sheared_level(lvl))
        else:
            assert torch.equal(a.corr_pyramid[lvl], want.corr_pyramid[lvl])

    # no free slot left: the stores grow (one copy), slot numbers stay valid
    big = CorrBlock(fm[ii[:8]][None], fm[jj[:8]][None], layout=layout)
    a = a.cat(big)
    sel = torch.cat([sel, torch.arange(8, device="cuda")])
    assert a.n == 13 and a.stats["grown"] == 1 and a.capacity >= 13
    c13 = _coords(13, h, w, 6)
    assert torch.equal(a(c13), fresh(sel)(c13))

    # a block that was already BUILT is copied in (its edges' bytes only).  (One lookup of a fresh small block does not build it --
    # dba_corr_build_lookup_once_sheared, test_gpu_corr.py -- the second one does.)
    used = CorrBlock(fm[ii[:2]][None], fm[jj[:2]][None], layout=layout)
    used(_coords(2, h, w, 7))
    used(_coords(2, h, w, 7))
    assert used._pending is None
    cap = a.capacity
    a = a[torch.arange(10, device="cuda")].cat(used)
    sel = torch.cat([sel[:10], torch.arange(2, device="cuda")])
    assert a.stats["copied_edges"] == 2 and a.capacity == cap
    c12 = _coords(12, h, w, 8)
    assert torch.equal(a(c12), fresh(sel)(c12))


def test_build_after_an_in_place_write_of_the_maps_raises():
    from dbaf_amd.corr import CorrBlock
    fm = _fmaps(3, 32, 16, 16, 2)
    f1, f2 = fm[:2][None].clone(), fm[1:][None].clone()
    blk = CorrBlock(f1, f2)
    f1.mul_(2.0)
    with pytest.raises(RuntimeError):
        blk(_coords(2, 16, 16, 0))


@pytest.mark.parametrize("h,w", [(64, 64), (20, 64), (28, 107), (16, 16), (12, 128), (8, 192)])
def test_lookup_with_the_reprojection_in_its_launch(h, w, lookup_kernel):
    """one launch == dba_reproject + lookup, bit for bit (coordinates, validity, correlation features): maps whose rows
    fill whole waves (streaming form: the edge geometry is shared through LDS, workgroups that straddle two edges at h = 20),
    maps that do not (resident form), a stereo edge (ii == jj), with and without a slot table"""
    from dbaf_amd.corr import CorrBlock
    from dbaf_amd import projective_ops as pops
    nkf = 6
    gi, gj = syn.graph_banded(nkf, 2)
    gi, gj = np.concatenate([gi, [2]]), np.concatenate([gj, [2]])      # + a stereo edge
    W = syn.make_window(gi, gj, nkf, h, w, seed=3, intr=(0.37 * w, 0.37 * w, 0.5 * w - 0.3, 0.5 * h + 0.2))
    fm = _fmaps(W.B, 32, h, w, 9)
    ii, jj = torch.from_numpy(W.ii).cuda(), torch.from_numpy(W.jj).cuda()
    poses, disps = torch.from_numpy(W.poses).cuda(), torch.from_numpy(W.disps).cuda()
    K = torch.from_numpy(np.tile(W.intrinsics, (W.B, 1))).cuda()
    n = len(W.ii)
    corr = CorrBlock(fm[ii][None], fm[jj][None])
    out, coords, valid = corr.lookup_reprojected(poses, disps, K, ii, jj)
    c2, v2 = pops.projective_transform(poses[None], disps[None], K[None], ii, jj)
    assert torch.equal(coords, c2) and torch.equal(valid, v2)
    assert torch.equal(out, corr(c2))
    assert out.shape == (1, n, 196, h, w)
    # intrinsics as [4] (shared), poses as [1, B, 7]
    out_b, coords_b, _ = corr.lookup_reprojected(poses[None], disps[None], K[0], ii, jj)
    assert torch.equal(out_b, out) and torch.equal(coords_b, coords)
    # behind a slot table: drop every third edge, add two back
    keep = torch.ones(n, dtype=torch.bool, device="cuda")
    keep[::3] = False
    corr = corr[keep].cat(CorrBlock(fm[ii[:2]][None], fm[jj[:2]][None]))
    ii2, jj2 = torch.cat([ii[keep], ii[:2]]), torch.cat([jj[keep], jj[:2]])
    out2, coords2, valid2 = corr.lookup_reprojected(poses, disps, K, ii2, jj2)
    c3, v3 = pops.projective_transform(poses[None], disps[None], K[None], ii2, jj2)
    assert torch.equal(coords2, c3) and torch.equal(valid2, v3) and torch.equal(out2, corr(c3))
    sel = torch.cat([torch.nonzero(keep).view(-1), torch.arange(2, device="cuda")])
    assert torch.equal(out2, out[:, sel])


def test_zero_edit_shadows_match_the_edges_of_new_tensors():
    """droid_backends.corr_index_forward under DBA_ZERO_EDIT_SHADOW_MATCH: the level tensors a graph change creates
    (torch.cat / boolean index of the old ones, /root/reference/dbaf/modules/corr.py:52-60) find the shadows of the edges
    they kept; only unseen edges are re-laid out; an in-place write is never matched; results equal the direct kernel"""
    import droid_backends
    from droid_backends import _SHADOWS
    from dbaf_amd.corr import CorrBlock
    h, w = 24, 40
    fm = _fmaps(6, 32, h, w, 12)
    ii = torch.tensor([0, 1, 1, 2, 2, 3, 3, 4, 4, 5], device="cuda")
    jj = torch.tensor([1, 0, 2, 1, 3, 2, 4, 3, 5, 4], device="cuda")
    pyr = CorrBlock.build_pyramid(fm[ii][None], fm[jj][None], 4)
    saved = (_SHADOWS.enabled, _SHADOWS.match, _SHADOWS._min_uses)

    def look(levels, c):
        cp = c[0].permute(0, 3, 1, 2).contiguous()
        return [droid_backends.corr_index_forward(levels[l], cp / 2 ** l, 3)[0] for l in range(4)]

    def direct(levels, c):
        _SHADOWS.enabled = False
        try:
            return look(levels, c)
        finally:
            _SHADOWS.enabled = True

    try:
        _SHADOWS.enabled, _SHADOWS.match = True, True
        _SHADOWS.min_uses = 2
        _SHADOWS.clear()
        a = [p[:7].contiguous() for p in pyr]
        c7 = _coords(7, h, w, 1)
        want = direct(a, c7)
        e0 = _SHADOWS.built_edges
        for _ in range(3):
            for g, r in zip(look(a, c7), want):
                assert torch.equal(g, r)
        assert _SHADOWS.built_edges == e0 + 4 * 7
        # add_factors: cat of three new edges -> new tensors; their second lookup re-lays out 3 edges per level, not 10
        b = [torch.cat([x, p[7:]], 0) for x, p in zip(a, pyr)]
        del a
        c10 = _coords(10, h, w, 2)
        want = direct(b, c10)
        e1, m1 = _SHADOWS.built_edges, _SHADOWS.matched_edges
        for _ in range(3):
            for g, r in zip(look(b, c10), want):
                assert torch.equal(g, r)
        assert _SHADOWS.built_edges == e1 + 4 * 3 and _SHADOWS.matched_edges == m1 + 4 * 7
        # rm_factors: boolean index -> new tensors, every edge known
        mask = torch.tensor([1, 0, 1, 1, 0, 1, 1, 1, 0, 1], dtype=torch.bool, device="cuda")
        cc = [x[mask] for x in b]
        del b
        c7b = _coords(7, h, w, 3)
        want = direct(cc, c7b)
        e2 = _SHADOWS.built_edges
        for _ in range(2):
            for g, r in zip(look(cc, c7b), want):
                assert torch.equal(g, r)
        assert _SHADOWS.built_edges == e2
        # an in-place write: the tensor's entry is dropped, nothing is matched, the new content is what comes back
        cc[0][2].mul_(0.5)
        want = direct(cc, c7b)
        for _ in range(3):
            for g, r in zip(look(cc, c7b), want):
                assert torch.equal(g, r)
        assert _SHADOWS.built_edges == e2 + 7
    finally:
        _SHADOWS.enabled, _SHADOWS.match, _SHADOWS._min_uses = saved
