"""The update-call dump (dbaf_amd/replay.py): schema round trip on the CPU, and the committed dump replayed on the GPU through
tools/replay_dump.py -- the path a recorded TUM-VI `graph.update()` call takes."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DUMP = os.path.join(HERE, "golden", "update_call_tiny_b.npz")


def test_dump_schema_round_trip(tmp_path):
    from dbaf_amd import synthetic as syn
    from dbaf_amd.replay import dump_update_call, load_update_call, SCHEMA
    W = syn.window_tiny_a(3)
    p = dump_update_call(str(tmp_path / "call.npz"), W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta,
                         W.ii, W.jj, W.t0, W.t1, 2, W.lm, W.ep, False)
    L = load_update_call(p)
    for k in ("poses", "disps", "intrinsics", "disps_sens", "target", "weight", "eta", "ii", "jj"):
        assert np.array_equal(getattr(L, k), getattr(W, k)), k
    assert (L.t0, L.t1, L.itrs, L.N, L.M) == (W.t0, W.t1, 2, W.N, W.M) and L.fmaps is None
    assert abs(L.lm - W.lm) < 1e-10 and abs(L.ep - W.ep) < 1e-7   # (stored as float32, as the binding takes them)
    committed = load_update_call(DUMP)
    assert committed.N == 14 and committed.fmaps is not None and committed.coords.shape == (14, 24, 32, 2)
    with np.load(DUMP) as z:
        assert set(SCHEMA) <= set(z.files)
    # a dump that lacks a key is refused
    with np.load(p) as z:
        rec = {k: z[k] for k in z.files if k != "eta"}
    np.savez(str(tmp_path / "broken.npz"), **rec)
    with pytest.raises(ValueError):
        load_update_call(str(tmp_path / "broken.npz"))


@pytest.mark.gpu
def test_committed_dump_replays_on_the_device():
    sys.path.insert(0, os.path.join(HERE, "..", "tools"))
    import replay_dump
    out = replay_dump.replay(DUMP, verbose=False)
    assert out["N"] == 14 and out["lookup_bit_exact"] and "dt=" in out["ba"]
