"""CPU: tests/golden/caller_dumps.npz -- every `droid_backends` call the reference's OWN caller made
(CovisibleGraph.add_factors / update / rm_factors over DepthVideo, with the reference's UpdateModule; recorded by
tests/golden/make_caller_dumps.py, which imports /root/reference in the authoring container; data only).

Here: the oracle reproduces what was recorded (the recorder's ops WERE the oracle, so this pins the oracle build of the
test box to the one of the authoring container), and the layouts the call sites build are what the C ABI documents:
corr [B,N,196,h,w] with channel = level * 49 + x_offset * 7 + y_offset (dbaf/modules/corr.py:40-50), motion [B,N,4,h,w],
delta / weight [B,N,h,w,2] (dbaf/droid_net.py:117-142), targets / weights [N,2,h,w] (covisible_graph.py:332-333), eta
(:330), the inactive edges in front (:242-247).  The GPU half is tests/test_gpu_reference_caller.py."""
import hashlib
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DUMPS = os.path.join(HERE, "golden", "caller_dumps.npz")


class CallerDumps:
    """read access to the fixture: calls in order, the pyramid states they refer to"""

    def __init__(self, path=DUMPS):
        self.z = np.load(path)
        self.kinds = [str(k) for k in self.z["kinds"]]
        self.marks = dict(zip((str(s) for s in self.z["marks_label"]), (int(c) for c in self.z["marks_count"])))
        self.pyr12 = [self.z["pyramid12_lvl%d" % l] for l in range(4)]
        self.keep = self.z["keep_idx_after_rm"]

    def call(self, k):
        pre = "call%03d_" % k
        return {f[len(pre):]: self.z[f] for f in self.z.files if f.startswith(pre)}

    def calls(self, kind):
        return [(k, self.call(k)) for k, kd in enumerate(self.kinds) if kd == kind]

    def volume(self, c):
        """the level tensor a recorded corr_index_forward call was handed"""
        lvl, st = int(c["lvl"]), int(c["volume_state"])
        v = self.pyr12[lvl]
        return v if st == 0 else (v[:8] if st == 1 else v[self.keep])

    def updates(self):
        """per recorded update(): (its four lookup calls, its ba call)"""
        out, look = [], []
        for k, kd in enumerate(self.kinds):
            if kd == "corr_index_forward":
                look.append(self.call(k))
            elif kd == "ba":
                out.append((look, self.call(k)))
                look = []
        return out


@pytest.fixture(scope="module")
def dumps():
    return CallerDumps()


def _oracle():
    from oracle import oracle as orc
    return orc


def test_fixture_holds_the_recorded_sequence(dumps):
    assert dumps.kinds.count("corr_index_forward") == 20 and dumps.kinds.count("ba") == 5
    assert dumps.kinds.count("frame_distance") == 2
    assert list(dumps.marks) == ["add_factors_8", "update_0", "update_1", "add_factors_4", "update_2", "rm_factors", "update_3",
                                 "update_4_motion_only", "add_proximity_factors"]
    ups = dumps.updates()
    assert [len(l) for l, _ in ups] == [4] * 5
    # what the call sites hand over: whole buffers, per-frame damping rows, [N,2,h,w] targets, int64 edges
    _, b = ups[0]
    assert b["poses"].shape == (12, 7) and b["disps"].shape == (12, 16, 16) and b["intrinsics"].shape == (4,)
    assert b["target"].shape == (8, 2, 16, 16) and b["weight"].shape == (8, 2, 16, 16) and b["ii"].dtype == np.int64
    assert (int(b["t0"]), int(b["t1"]), int(b["itrs"])) == (1, 4, 2)
    # eta = .2 * damping[unique(ii)] + EP, one row per source frame (covisible_graph.py:330)
    assert b["eta"].shape == (len(np.unique(b["ii"])), 16, 16) and np.allclose(b["eta"], 0.2 * 1e-6 + 1e-7)
    # the update after rm_factors(store=True) runs with the inactive edges IN FRONT (covisible_graph.py:242-247) and a
    # later t0: poses 0, 1 are fixed there, their edges still constrain the window
    _, b3 = ups[3]
    assert int(b3["t0"]) == 2 and list(b3["ii"][:4]) == [0, 1, 0, 2] and list(b3["jj"][:4]) == [1, 0, 2, 0]
    _, b4 = ups[4]
    assert bool(b4["motion_only"]) and int(b4["itrs"]) == 3 and np.array_equal(b4["disps"], b4["disps_out"])


def test_oracle_reproduces_the_recorded_lookups_bit_for_bit(dumps):
    orc = _oracle()
    for k, c in dumps.calls("corr_index_forward"):
        out = orc.corr_index_forward(dumps.volume(c), c["coords"], 3)
        assert hashlib.sha256(out.tobytes()).hexdigest() == str(c["out_sha256"]), "lookup call %d" % k


def test_update_operator_layouts(dumps):
    """the tensors around the update operator, as the reference's own modules produced / consumed them"""
    orc = _oracle()
    z = dumps.z
    look, b = dumps.updates()[0]
    n, h, w = 8, 16, 16
    assert z["upd_corr"].shape == (1, n, 196, h, w) and z["upd_motion"].shape == (1, n, 4, h, w)
    assert z["upd_delta"].shape == (1, n, h, w, 2) and z["upd_weight"].shape == (1, n, h, w, 2)
    # corr = cat over levels of corr_index_forward(level l, coords / 2^l).view(1, n, 49, h, w): channel = l*49 + a*7 + b
    for lvl, c in enumerate(look):
        assert int(c["lvl"]) == lvl
        if lvl:
            assert np.array_equal(c["coords"], look[0]["coords"] / np.float32(2 ** lvl))    # corr.py:47
        out = orc.corr_index_forward(dumps.volume(c), c["coords"], 3)                        # [n, 7, 7, h, w]
        assert np.array_equal(out.reshape(n, 49, h, w).view(np.uint16),
                              z["upd_corr"][0, :, 49 * lvl:49 * lvl + 49].view(np.uint16))
    # the fused lookup of this repo writes that very tensor from [n,h,w,2] coordinates
    coords_nhw2 = np.ascontiguousarray(look[0]["coords"].transpose(0, 2, 3, 1))
    fused = orc.corr_lookup_pyramid([dumps.volume(c) for c in look], coords_nhw2, 3)
    assert np.array_equal(fused.view(np.uint16), z["upd_corr"][0].view(np.uint16))
    # motion = cat(coords1 - coords0, target - coords1) as [1,n,4,h,w], clamped (covisible_graph.py:220-222); at the first
    # update target is still the reprojection add_factors stored, and nothing moved since: the second half is zero
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    flow = look[0]["coords"] - np.stack([xx, yy])[None]
    assert np.allclose(z["upd_motion"][0, :, :2], np.clip(flow, -64, 64), atol=1e-6)
    assert np.abs(z["upd_motion"][0, :, 2:]).max() == 0.0
    # target = coords1 + delta, weight: [1,n,h,w,2] -> view(-1,h,w,2).permute(0,3,1,2): channel 0 = x (:332-333)
    tgt = look[0]["coords"] + z["upd_delta"][0].transpose(0, 3, 1, 2)
    assert np.allclose(b["target"], tgt, atol=1e-6)
    wt = z["upd_weight"][0].transpose(0, 3, 1, 2).copy()
    wt[b["ii"] == b["ii"].max()] /= 10.0      # newest-frame down-weighting (:323-326)
    wt[b["jj"] == b["jj"].max()] /= 4.0
    assert np.allclose(b["weight"], wt, rtol=1e-6, atol=1e-9)


def test_oracle_reproduces_the_recorded_ba_and_distances(dumps):
    orc = _oracle()
    for k, b in dumps.calls("ba"):
        r = orc.ba(b["poses"], b["disps"], b["intrinsics"], b["disps_sens"], b["target"], b["weight"], b["eta"], b["ii"],
                   b["jj"], int(b["t0"]), int(b["t1"]), int(b["itrs"]), float(b["lm"]), float(b["ep"]),
                   bool(b["motion_only"]), 0.05, np.float32)
        # (same C source, possibly another compiler / core count: fp32 reassociation in the OpenMP reductions)
        np.testing.assert_allclose(r["poses"], b["poses_out"], rtol=0, atol=2e-6, err_msg="ba call %d" % k)
        np.testing.assert_allclose(r["disps"], b["disps_out"], rtol=2e-5, atol=2e-6, err_msg="ba call %d" % k)
    for k, f in dumps.calls("frame_distance"):
        d = orc.frame_distance(f["poses"], f["disps"], f["intrinsics"], f["ii"], f["jj"], float(f["beta"]), np.float32)
        np.testing.assert_allclose(d, f["out"], rtol=1e-5, atol=1e-6)


def test_reference_reprojection_through_the_se3_shim_matches_the_oracle(dumps):
    """the coordinates the lookups were handed came out of the reference's pops.projective_transform running on this repo's
    lietorch shim (DepthVideo.reproject, depth_video.py:221-229): the C oracle's reprojection must agree"""
    orc = _oracle()
    for look, b in dumps.updates():
        n_act = look[0]["coords"].shape[0]
        ii, jj = b["ii"][-n_act:], b["jj"][-n_act:]            # the active edges (inactive ones are in front)
        K = np.tile(b["intrinsics"], (b["poses"].shape[0], 1))
        c, _ = orc.reproject(b["poses"], b["disps"], K, ii, jj, np.float64)
        np.testing.assert_allclose(c, look[0]["coords"].transpose(0, 2, 3, 1), rtol=1e-5, atol=2e-4)
