#!/usr/bin/env python
"""Replay a recorded `graph.update()` call (dbaf_amd/replay.py schema) through the HIP path and the CPU oracle:

    python tools/replay_dump.py dump.npz [more.npz ...]

Per dump: droid_backends.ba on the device against the oracle's float64 arbiter and its fp32-faithful restatement of the
reference (poses: m / rad; inverse depths: relative, the criteria of tests/util.py::check_state), and -- when the dump
carries feature maps and coordinates -- the 4-level correlation lookup, bit for bit against the oracle's.
Test infrastructure (imports oracle/): not part of the product path."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "dba-fusion_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def replay(path, verbose=True):
    import torch
    import droid_backends
    from dbaf_amd.replay import load_update_call
    from dbaf_amd.corr import CorrBlock
    from oracle import oracle as orc
    from util import check_state
    W = load_update_call(path)
    dev = "cuda"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    out = dict(path=path, N=W.N, keyframes=W.t1, map=[W.h, W.w])
    args = (W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta, W.ii, W.jj, W.t0, W.t1, W.itrs, W.lm,
            W.ep, W.motion_only, 0.05)
    r64 = orc.ba(*args, np.float64)
    r32 = orc.ba(*args, np.float32)
    poses, disps = t(W.poses), t(W.disps)
    droid_backends.ba(poses, disps, t(W.intrinsics), t(W.disps_sens), t(W.target), t(W.weight), t(W.eta), t(W.ii), t(W.jj),
                      W.t0, W.t1, W.itrs, W.lm, W.ep, W.motion_only)
    torch.cuda.synchronize()
    clamp = lambda a: np.maximum(a, 0.001)  # noqa: E731  (depth_video.py:560)
    out["ba"] = check_state(poses.cpu().numpy(), clamp(disps.cpu().numpy()), r64["poses"], clamp(r64["disps"]), W.disps,
                            ref32_disps=clamp(r32["disps"]), ref32_poses=r32["poses"])
    if W.fmaps is not None and W.coords is not None:
        fm = t(W.fmaps)
        ii, jj = t(W.ii), t(W.jj)
        cb = CorrBlock(fm[ii][None], fm[jj][None], num_levels=4, radius=3)
        got = cb(t(W.coords)[None])[0].cpu().numpy()
        ref_pyr = [p.cpu().numpy() for p in CorrBlock.build_pyramid(fm[ii][None], fm[jj][None], 4)]
        ref = orc.corr_lookup_pyramid(ref_pyr, W.coords, 3)
        out["lookup_bit_exact"] = bool(np.array_equal(got.view(np.uint16), ref.view(np.uint16)))
        assert out["lookup_bit_exact"], "lookup differs from the oracle"
    if verbose:
        print(out)
    return out


if __name__ == "__main__":
    for pth in sys.argv[1:]:
        replay(pth)
