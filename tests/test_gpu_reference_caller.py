"""GPU: the calls the reference's OWN caller made (tests/golden/caller_dumps.npz: CovisibleGraph.add_factors / update /
rm_factors over DepthVideo with the reference's UpdateModule, recorded in the authoring container by
tests/golden/make_caller_dumps.py) replayed through the HIP path.

  * every recorded `corr_index_forward` call through droid_backends (both the direct kernel and the flow-aligned shadow it
    builds at the second use): bit-identical to what the caller received;
  * the same lookups through the slot-addressed CorrBlock driven like the caller drove its own (8 edges, cat of 4 more,
    boolean index after rm_factors), plain and with the reprojection in the launch: bit-identical;
  * the volumes themselves (MFMA build from the recorded feature maps) against the recorded pyramid;
  * every recorded `ba` call at the north-star tolerance against the float64 arbiter, with the recorded fp32 results (the
    reference's arithmetic, restated) as the second yardstick; `frame_distance`; the reprojection against the coordinates
    the reference's torch path produced on this repo's SE3 shim."""
import hashlib

import numpy as np
import pytest
import torch

from test_reference_caller_fixture import CallerDumps
from util import check_state

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle as orc
    return orc


@pytest.fixture(scope="module")
def dumps():
    return CallerDumps()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()


@pytest.mark.parametrize("policy", ["direct", "shadow_at_second_use", "match"])
def test_recorded_corr_index_forward_calls_bit_identical(dumps, policy):
    """the 20 recorded calls in order, on one tensor object per (graph state, level) like the caller holds them: through
    the direct kernel, through shadows built at a tensor's second use, and with the edges of the tensors the graph changes
    created (cat of 4 edges, boolean index) matched to the shadows already held"""
    import droid_backends
    from droid_backends import _SHADOWS
    saved = (_SHADOWS.enabled, _SHADOWS.match, _SHADOWS._min_uses)
    _SHADOWS.enabled, _SHADOWS.match = policy != "direct", policy == "match"
    _SHADOWS.min_uses = 2
    _SHADOWS.clear()
    try:
        vols = {}
        b0, e0, m0 = _SHADOWS.builds, _SHADOWS.built_edges, _SHADOWS.matched_edges
        for k, c in dumps.calls("corr_index_forward"):
            key = (int(c["volume_state"]), int(c["lvl"]))
            if key not in vols:
                vols[key] = _t(dumps.volume(c))
            out, = droid_backends.corr_index_forward(vols[key], _t(c["coords"]), 3)
            assert out.shape == (vols[key].shape[0], 7, 7, 16, 16) and out.dtype == torch.float16
            assert _sha(out) == str(c["out_sha256"]), "recorded lookup call %d (%s)" % (k, policy)
        if policy == "shadow_at_second_use":
            # states 1 (8 edges: updates 0, 1) and 2 (8 edges: updates 3, 4) are looked up twice per level, state 0 once
            assert _SHADOWS.builds == b0 + 8 and _SHADOWS.built_edges == e0 + 64
        if policy == "match":
            # state 1 (8 edges) gets its shadows at its second lookup; state 0's tensors (the cat: those 8 + 4 new edges, a
            # third of them unknown) are looked up once and go to the direct kernel; state 2's (the boolean index: 4 of
            # state 1's edges + the 4 never re-laid out) match 4 and re-lay out 4 per level at their second lookup
            assert _SHADOWS.built_edges == e0 + 4 * (8 + 4) and _SHADOWS.matched_edges == m0 + 4 * 4
    finally:
        _SHADOWS.enabled, _SHADOWS.match, _SHADOWS._min_uses = saved


@pytest.mark.parametrize("layout", ["sheared", "reference"])
def test_slot_addressed_corrblock_driven_like_the_recorded_caller(dumps, layout, lookup_kernel):
    from dbaf_amd.corr import CorrBlock
    z = dumps.z
    p12 = [_t(p) for p in dumps.pyr12]
    keep = torch.from_numpy(dumps.keep).cuda()

    def block(levels):
        return CorrBlock.from_reference(levels) if layout == "sheared" else CorrBlock.from_pyramid(levels, "reference")

    ups = dumps.updates()
    # add_factors(8): a block of 8 edges with room for the 4 to come
    corr = block([p[:8].contiguous() for p in p12])
    corr._grow(12)
    assert corr.capacity >= 12

    def lookup(u):
        look, _ = ups[u]
        c = _t(np.ascontiguousarray(look[0]["coords"].transpose(0, 2, 3, 1)))[None]
        got = corr(c)[0]
        for lvl, call in enumerate(look):
            assert _sha(got[:, 49 * lvl:49 * lvl + 49].reshape(-1, 7, 7, 16, 16).contiguous()) == str(call["out_sha256"]), \
                "update %d level %d" % (u, lvl)
        return got

    got0 = lookup(0)
    assert np.array_equal(got0.cpu().numpy().view(np.uint16), z["upd_corr"][0].view(np.uint16))   # what the GRU consumed
    lookup(1)
    corr = corr.cat(block([p[8:].contiguous() for p in p12]))          # add_factors(4): covisible_graph.py:131
    assert corr.n == 12 and corr.stats["grown"] == 1                    # (the one grow above; the cat found its slots)
    lookup(2)
    mask = torch.ones(12, dtype=torch.bool, device="cuda")
    mask[keep] = False
    corr = corr[~mask]                                                  # rm_factors: covisible_graph.py:166
    assert corr.n == len(dumps.keep)
    lookup(3)
    lookup(4)
    # the slots of the dropped edges are free again: four new edges land there without growing
    grown = corr.stats["grown"]
    corr = corr.cat(block([p[:4].contiguous() for p in p12]))
    assert corr.n == 12 and corr.stats["grown"] == grown
    want = block([torch.cat([p[keep], p[:4]], 0) for p in p12])     # what the reference's cat / index would hold now
    c = _t(np.ascontiguousarray(ups[2][0][0]["coords"].transpose(0, 2, 3, 1)))[None]
    assert torch.equal(corr(c), want(c))
    if layout == "reference":
        for have, ref in zip(corr.corr_pyramid, want.corr_pyramid):
            assert torch.equal(have, ref)


def test_fused_reprojection_lookup_on_the_recorded_states(dumps, lookup_kernel):
    """reprojection + lookup in one launch on the recorded video state == dba_reproject followed by the lookup, bit for
    bit; and the coordinates agree with what the reference's torch / lietorch path handed its own lookups"""
    from dbaf_amd.corr import CorrBlock
    from dbaf_amd import projective_ops as pops
    p12 = [_t(p) for p in dumps.pyr12]
    for u, (look, b) in enumerate(dumps.updates()):
        st = int(look[0]["volume_state"])
        sel = slice(0, 12) if st == 0 else (slice(0, 8) if st == 1 else torch.from_numpy(dumps.keep).cuda())
        corr = CorrBlock.from_reference([p[sel].contiguous() for p in p12])
        n = corr.n
        ii, jj = _t(b["ii"][-n:]), _t(b["jj"][-n:])
        poses, disps = _t(b["poses"]), _t(b["disps"])
        K = _t(np.tile(b["intrinsics"], (poses.shape[0], 1)))
        out, coords, valid = corr.lookup_reprojected(poses, disps, K, ii, jj)
        c2, v2 = pops.projective_transform(poses[None], disps[None], K[None], ii, jj)
        assert torch.equal(coords, c2) and torch.equal(valid, v2)
        assert torch.equal(out, corr(c2))
        ref = look[0]["coords"].transpose(0, 2, 3, 1)
        np.testing.assert_allclose(coords[0].cpu().numpy(), ref, rtol=1e-5, atol=2e-4)
        # where the recorded coordinates and ours agree to the bit, so do the lookups the caller received
        same = (coords[0].cpu().numpy() == ref).all(-1)                      # [n, h, w]
        got = out[0].cpu().numpy().reshape(n, 4, 49, 16, 16)
        orc = _oracle()
        for lvl, call in enumerate(look):
            want = orc.corr_index_forward(dumps.volume(call), call["coords"], 3).reshape(n, 49, 16, 16)
            m = np.broadcast_to(same[:, None], want.shape)
            assert np.array_equal(got[:, lvl][m].view(np.uint16), want[m].view(np.uint16))
        # and on OUR coordinates the oracle's lookup of the recorded volumes is what the launch wrote, everywhere
        want = orc.corr_lookup_pyramid([dumps.volume(call) for call in look], coords[0].cpu().numpy(), 3)
        assert np.array_equal(out[0].cpu().numpy().view(np.uint16), want.view(np.uint16))


def test_mfma_volume_build_matches_the_recorded_pyramid(dumps):
    from dbaf_amd.corr import CorrBlock
    fm = _t(dumps.z["fmaps"])
    _, b = dumps.updates()[2]
    ii, jj = _t(b["ii"]), _t(b["jj"])
    pyr = [p.cpu().numpy() for p in CorrBlock.build_pyramid(fm[ii][None], fm[jj][None], 4)]
    # level 0: the matrix cores' float accumulation order differs from any sequential one -> identical to the recorded
    # volume but for rare ties of the final rounding (one fp16 ulp)
    ref = dumps.pyr12[0]
    ulp = np.maximum(np.spacing(np.abs(ref)).astype(np.float64), 2.0 ** -24)
    d = np.abs(pyr[0].astype(np.float64) - ref.astype(np.float64)) / ulp
    assert (d > 0).mean() < 0.01 and d[np.abs(ref.astype(np.float32)) > 0.01].max() <= 1.0
    # levels 1-3: the 2x2 average of the ROUNDED level below (F.avg_pool2d on half, corr.py:38), bit for bit
    orc = _oracle()
    for lvl in range(1, 4):
        assert np.array_equal(orc.avg_pool2(pyr[lvl - 1]).view(np.uint16), pyr[lvl].view(np.uint16)), lvl
    # and a block built from the maps answers the recorded lookups to the last bit wherever its volume equals the recorded one
    blk = CorrBlock(fm[ii][None], fm[jj][None])
    look, _ = dumps.updates()[2]
    c = _t(np.ascontiguousarray(look[0]["coords"].transpose(0, 2, 3, 1)))[None]
    got = blk(c)[0].float().cpu().numpy()
    want = _oracle().corr_lookup_pyramid(dumps.pyr12, c[0].cpu().numpy(), 3).astype(np.float32)
    assert np.abs(got - want).max() <= 2.0 ** -8 * max(1.0, np.abs(want).max())


def test_recorded_ba_calls_at_the_north_star_tolerance(dumps):
    import droid_backends
    orc = _oracle()
    for k, b in dumps.calls("ba"):
        args = (b["poses"], b["disps"], b["intrinsics"], b["disps_sens"], b["target"], b["weight"], b["eta"], b["ii"], b["jj"],
                int(b["t0"]), int(b["t1"]), int(b["itrs"]), float(b["lm"]), float(b["ep"]), bool(b["motion_only"]))
        r64 = orc.ba(*args, 0.05, np.float64)
        poses, disps = _t(b["poses"]), _t(b["disps"])
        droid_backends.ba(poses, disps, _t(b["intrinsics"]), _t(b["disps_sens"]), _t(b["target"]), _t(b["weight"]),
                          _t(b["eta"]), _t(b["ii"]), _t(b["jj"]), *args[9:])
        disps.clamp_(min=0.001)                                   # depth_video.py:560
        clamp = lambda a: np.maximum(a, 0.001)                    # noqa: E731
        # north_star's bound as written (1e-4 of the depth, 1e-5 m / 1e-6 rad), no allowance; the recorded outputs -- the
        # reference's fp32 arithmetic (restated) on the authoring box -- are logged as the second yardstick (measured 2-4e-6)
        check_state(poses.cpu().numpy(), disps.cpu().numpy(), r64["poses"], clamp(r64["disps"]), b["disps"],
                    ref32_disps=None, d_rtol=1e-4, log32_disps=clamp(b["disps_out"]), log32_poses=b["poses_out"])
        if bool(b["motion_only"]):
            assert torch.equal(disps, _t(b["disps"]).clamp(min=0.001))


def test_recorded_frame_distance_calls(dumps):
    import droid_backends
    for _, f in dumps.calls("frame_distance"):
        d = droid_backends.frame_distance(_t(f["poses"]), _t(f["disps"]), _t(f["intrinsics"]), _t(f["ii"]), _t(f["jj"]),
                                          float(f["beta"]))
        np.testing.assert_allclose(d.cpu().numpy(), f["out"], rtol=2e-4, atol=1e-5)
