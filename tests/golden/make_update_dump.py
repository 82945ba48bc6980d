"""Writes tests/golden/update_call_tiny_b.npz: one graph.update() call in the dump schema of dbaf_amd/replay.py, produced
from the synthetic 6-keyframe / 14-edge / 24x32 window (stereo edge, fixed-pose edges, sensor depths), with 32-channel
feature maps and the lookup coordinates.  No recorded dataset exists on the build boxes; the GPU test replays this file
through the same loader and driver a recorded TUM-VI call would take (tools/replay_dump.py).
    python tests/golden/make_update_dump.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "dba-fusion_amd"))
from dbaf_amd import synthetic as syn  # noqa: E402
from dbaf_amd.replay import dump_update_call  # noqa: E402

W = syn.window_tiny_b(5, with_fmaps=True, channels=32)
coords = syn.lookup_coords(W, oob_frac=0.05, seed=5)
dump_update_call(os.path.join(HERE, "update_call_tiny_b.npz"), W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight,
                 W.eta, W.ii, W.jj, W.t0, W.t1, 2, W.lm, W.ep, False, fmaps=W.fmaps, coords=coords)
print("written", os.path.getsize(os.path.join(HERE, "update_call_tiny_b.npz")), "bytes")
