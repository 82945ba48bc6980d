"""Dump / load of ONE `graph.update()` call's inputs, for replaying recorded real data through the HIP path and the oracle.

The reference holds these tensors at /root/reference/dbaf/covisible_graph.py:330-337, right before `self.video.ba(...)`:
    poses, disps, intrinsics, disps_sens   (DepthVideo buffers, whole arrays)
    target, weight                         (update-operator outputs, [N, 2, h, w] as handed to droid_backends.ba)
    eta                                    (damping, [|kx|, h, w] or [1, h, w])
    ii, jj, t0, t1, itrs, lm, ep, motion_only
and, for the lookup half of the update, optionally the keyframe feature maps `fmaps` [B, C, h, w] (fp16) and the
coordinates `coords` [N, h, w, 2] the correlation block was queried with.

A maintainer records a call with three lines in CovisibleGraph.update():

    from dbaf_amd.replay import dump_update_call
    dump_update_call("/tmp/update_%05d.npz" % k, self.video.poses, self.video.disps, self.video.intrinsics[0],
                     self.video.disps_sens, target, weight, damping, ii, jj, t0, t1, itrs, 1e-4, 0.1, motion_only,
                     fmaps=self.video.fmaps[:, 0], coords=coords1[0])

and replays it on the GPU box with `python tools/replay_dump.py /tmp/update_00012.npz` (HIP vs the CPU oracle, the parity
figures of tests/util.py).  The .npz holds plain arrays only.  No dataset exists on the build boxes, so the committed
example (tests/golden/update_call_tiny_b.npz) was written by this very function from dbaf_amd.synthetic.
"""
import numpy as np

SCHEMA = ("poses", "disps", "intrinsics", "disps_sens", "target", "weight", "eta", "ii", "jj", "t0", "t1", "itrs", "lm", "ep",
          "motion_only")
OPTIONAL = ("fmaps", "coords")
VERSION = 1


def _np(x, dtype=None):
    if hasattr(x, "detach"):
        x = x.detach().to("cpu").numpy()
    x = np.ascontiguousarray(x)
    return x.astype(dtype) if dtype is not None else x


def dump_update_call(path, poses, disps, intrinsics, disps_sens, target, weight, eta, ii, jj, t0, t1, itrs=2, lm=1e-4, ep=0.1,
                     motion_only=False, fmaps=None, coords=None):
    """torch tensors (any device) or arrays -> one compressed .npz following SCHEMA"""
    disps = _np(disps, np.float32)
    B, h, w = disps.shape
    rec = dict(schema_version=np.int32(VERSION), poses=_np(poses, np.float32).reshape(B, 7), disps=disps,
               intrinsics=_np(intrinsics, np.float32).reshape(-1)[:4], disps_sens=_np(disps_sens, np.float32).reshape(B, h, w),
               target=_np(target, np.float32), weight=_np(weight, np.float32), eta=_np(eta, np.float32).reshape(-1, h, w),
               ii=_np(ii, np.int64).reshape(-1), jj=_np(jj, np.int64).reshape(-1), t0=np.int32(t0), t1=np.int32(t1),
               itrs=np.int32(itrs), lm=np.float32(lm), ep=np.float32(ep), motion_only=np.bool_(motion_only))
    N = rec["ii"].shape[0]
    assert rec["target"].shape == (N, 2, h, w) and rec["weight"].shape == (N, 2, h, w), "target / weight must be [N,2,h,w]"
    if fmaps is not None:
        rec["fmaps"] = _np(fmaps, np.float16)
    if coords is not None:
        rec["coords"] = _np(coords, np.float32).reshape(N, h, w, 2)
    np.savez_compressed(path, **rec)
    return path


class UpdateCall:
    """the loaded dump, with the attribute names of dbaf_amd.synthetic's windows (so the same drivers take either)"""

    def __init__(self, rec):
        for k in SCHEMA:
            if k not in rec:
                raise ValueError("update-call dump lacks '%s'" % k)
        self.poses, self.disps, self.intrinsics = rec["poses"], rec["disps"], rec["intrinsics"]
        self.disps_sens, self.target, self.weight, self.eta = rec["disps_sens"], rec["target"], rec["weight"], rec["eta"]
        self.ii, self.jj = rec["ii"], rec["jj"]
        self.t0, self.t1, self.itrs = int(rec["t0"]), int(rec["t1"]), int(rec["itrs"])
        self.lm, self.ep, self.motion_only = float(rec["lm"]), float(rec["ep"]), bool(rec["motion_only"])
        self.B, self.h, self.w = self.disps.shape
        self.N = int(self.ii.shape[0])
        self.kx = np.unique(np.concatenate([np.arange(self.t0, self.t1), self.ii]))
        self.M = len(self.kx)
        self.fmaps = rec["fmaps"] if "fmaps" in rec else None
        self.coords = rec["coords"] if "coords" in rec else None


def load_update_call(path):
    with np.load(path) as z:
        rec = {k: z[k] for k in z.files}
    if int(rec.get("schema_version", 0)) != VERSION:
        raise ValueError("update-call dump: schema version %s, this loader reads %d" % (rec.get("schema_version"), VERSION))
    return UpdateCall(rec)
