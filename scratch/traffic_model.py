"""HBM read traffic of the fused lookup under different centrings of the sheared volume (numpy model, CPU).

The volume stores, for source pixel p and target (ty, tx) at level l, the line (dy, dx) = ((ty - cy_l(p)) mod h_l,
(tx - cx_l(p)) mod w_l); a line holds 64 consecutive pixels (128 B), HBM is fetched in 64-byte sectors = 32 pixels of one
line.  A sector is read when any of its pixels needs that (dy, dx).  Centres compared: the pixel's own position (what the
build writes today), and the integer reprojection at build time kept per pixel / per group of 4, 8, 64 pixels.
   python scratch/traffic_model.py [drift_px]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
from dbaf_amd import synthetic as syn  # noqa: E402

drift = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
W = syn.window_25_96()
coords, _ = syn.reproject_np(W.poses, W.disps, W.intrinsics, W.ii, W.jj)   # [N, h, w, 2] (x, y)
N, h, w = coords.shape[:3]
HW = h * w
rng = np.random.default_rng(0)
c0 = coords.reshape(N, HW, 2).astype(np.float64)
c_now = c0 + (drift * rng.standard_normal(c0.shape) if drift else 0.0)     # lookup-time coordinates
ys, xs = np.divmod(np.arange(HW), w)
r = 3


def sectors(level, centre_y, centre_x, e):
    hl, wl = h >> level, w >> level
    fy = np.floor(c_now[e, :, 1] / 2 ** level).astype(np.int64)
    fx = np.floor(c_now[e, :, 0] / 2 ** level).astype(np.int64)
    need = set()
    grp = np.arange(HW) >> 5
    o = np.arange(-r, r + 2)
    wy = fy[:, None] + o[None, :]
    wx = fx[:, None] + o[None, :]
    oky = (wy >= 0) & (wy < hl)
    okx = (wx >= 0) & (wx < wl)
    dy = np.mod(wy - centre_y[:, None], hl)
    dx = np.mod(wx - centre_x[:, None], wl)
    key = (grp[:, None, None] * hl + dy[:, :, None]) * wl + dx[:, None, :]
    ok = oky[:, :, None] & okx[:, None, :]
    return np.unique(key[ok]).size, int(ok.sum())


def centres(kind, level, e):
    if kind == "own":
        return ys >> level, xs >> level
    iy = np.floor(c0[e, :, 1]).astype(np.int64)
    ix = np.floor(c0[e, :, 0]).astype(np.int64)
    g = {"pixel": 1, "quad": 4, "oct": 8, "strip": 64}[kind]
    if g > 1:   # the group's offset (first pixel's reprojection minus its position) applied to every pixel's own position
        first = (np.arange(HW) // g) * g
        oy, ox = iy[first] - ys[first], ix[first] - xs[first]
        iy, ix = ys + oy, xs + ox
    return iy >> level, ix >> level


edges = range(0, N, 4)
print("drift %.2f px; 64-byte sectors read / (taps x 2 B / 64), summed over levels, %d edges" % (drift, len(list(edges))))
for kind in ("own", "strip", "oct", "quad", "pixel"):
    tot_s = tot_t = 0
    per_level = []
    for level in range(4):
        s_l = t_l = 0
        for e in edges:
            cy, cx = centres(kind, level, e)
            s, t = sectors(level, cy, cx, e)
            s_l += s
            t_l += t
        per_level.append(s_l * 64 / (t_l * 2.0))
        tot_s += s_l
        tot_t += t_l
    print("%-6s  %.3f   per level %s" % (kind, tot_s * 64 / (tot_t * 2.0), " ".join("%.3f" % v for v in per_level)))
