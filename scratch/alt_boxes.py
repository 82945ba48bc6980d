"""scratch (CPU): union box of the windows of a tile of source pixels on the bench scene, per level, for tile shapes --
sizes the MFMA form of the on-the-fly correlation has to cover.  python scratch/alt_boxes.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dba-fusion_amd"))
import numpy as np
from dbaf_amd import synthetic as syn
W = syn.window_25_96(0)
scene, _ = syn.reproject_np(W.poses, W.disps, W.intrinsics, W.ii, W.jj)   # [N, 64, 64, 2]
N = scene.shape[0]
for th, tw in ((4, 16), (8, 8), (4, 8), (2, 16)):
    for l in range(4):
        f = np.floor(scene / 2 ** l).astype(np.int64)
        ox, oy = f[..., 0] - 3, f[..., 1] - 3
        hl = 64 >> l
        hit = (ox + 8 > 0) & (ox < hl) & (oy + 8 > 0) & (oy < hl)
        g = lambda a: a.reshape(N, 64 // th, th, 64 // tw, tw).transpose(0, 1, 3, 2, 4).reshape(-1, th * tw)
        gx, gy, gh = g(ox), g(oy), g(hit)
        big = 1 << 20
        x0 = np.where(gh, gx, big).min(1); x1 = np.where(gh, gx, -big).max(1)
        y0 = np.where(gh, gy, big).min(1); y1 = np.where(gh, gy, -big).max(1)
        anyh = gh.any(1)
        area = np.where(anyh, (x1 - x0 + 8) * (y1 - y0 + 8), 0)
        # clipped to the map
        cx0, cx1 = np.maximum(x0, 0), np.minimum(x1 + 8, hl); cy0, cy1 = np.maximum(y0, 0), np.minimum(y1 + 8, hl)
        carea = np.where(anyh, np.maximum(cx1 - cx0, 0) * np.maximum(cy1 - cy0, 0), 0)
        pc = np.percentile(area, [50, 90, 99, 99.9, 100])
        pcc = np.percentile(carea, [50, 90, 99, 99.9, 100])
        print("%dx%-2d level %d: box area p50 %4d p90 %4d p99 %4d p99.9 %5d max %6d | clipped to map p50 %4d p90 %4d p99 %4d max %5d | > 384: %.4f > 512: %.4f (clipped %.4f %.4f)" % (
            th, tw, l, *pc, pcc[0], pcc[1], pcc[2], pcc[4], (area > 384).mean(), (area > 512).mean(), (carea > 384).mean(), (carea > 512).mean()))
