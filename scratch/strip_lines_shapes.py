"""scratch (CPU): lines per 64-pixel strip of the FLATTENED index on the non-64-wide bench windows: bounding box of the strip's
window origins (what the resident form stages) against the exact union of the windows, per level."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dba-fusion_amd"))
import numpy as np
from dbaf_amd import synthetic as syn
for name, W in (("32_122 28x107", syn.window_32_122(0)), ("25_96 64x64", syn.window_25_96(0)),
                ("9_36 55x55", syn.make_window(*syn.graph_banded(9, 2, extra=[(0, 3), (1, 4), (2, 5)]), 9, 55, 55, seed=14, intr=(20.5, 20.5, 27.4, 27.6)))):
    scene, _ = syn.reproject_np(W.poses, W.disps, W.intrinsics, W.ii, W.jj)
    N, h, w = scene.shape[:3]
    HW = h * w
    ns = (HW + 63) // 64
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    out = []
    for l in range(4):
        hl, wl = h >> l, w >> l
        f = np.floor(scene / 2 ** l).astype(np.int64)
        ox = (f[..., 0] - 3 - (xx >> l)[None]).reshape(N, HW)
        oy = (f[..., 1] - 3 - (yy >> l)[None]).reshape(N, HW)
        hit = ((f[..., 0] - 3 + 8 > 0) & (f[..., 0] - 3 < wl) & (f[..., 1] - 3 + 8 > 0) & (f[..., 1] - 3 < hl)).reshape(N, HW)
        box = exact = 0
        cnt = 0
        for e in range(0, N, max(1, N // 24)):
            for s in range(ns):
                sl = slice(s * 64, min(HW, s * 64 + 64))
                m = hit[e, sl]
                if not m.any():
                    continue
                gx, gy = ox[e, sl][m], oy[e, sl][m]
                x0, y0 = gx.min(), gy.min()
                nx, ny = gx.max() - x0 + 8, gy.max() - y0 + 8
                box += nx * ny
                if nx * ny <= 4096:
                    mm = np.zeros((ny, nx), bool)
                    for px, py in set(zip((gx - x0).tolist(), (gy - y0).tolist())):
                        mm[py:py + 8, px:px + 8] = True
                    exact += mm.sum()
                else:
                    exact += nx * ny
                cnt += 1
        out.append((box / cnt, exact / cnt))
    print("%-14s bounding box %s (mean %.1f) | exact union %s (mean %.1f)" % (
        name, " ".join("%6.1f" % b for b, _ in out), np.mean([b for b, _ in out]), " ".join("%6.1f" % x for _, x in out), np.mean([x for _, x in out])))
