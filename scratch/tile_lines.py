"""scratch (CPU): exact number of 128-byte lines a lookup reads on the bench scene per group of 64 pixels and level, for
different shapes of that group (the pixel order of the flow-aligned planes): the distinct (dy, dx) offsets over the
8 x 8 windows of the group's pixels.  python scratch/tile_lines.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dba-fusion_amd"))
import numpy as np
from dbaf_amd import synthetic as syn
W = syn.window_25_96(0)
scene, _ = syn.reproject_np(W.poses, W.disps, W.intrinsics, W.ii, W.jj)   # [N, 64, 64, 2]
N = scene.shape[0]
yy, xx = np.meshgrid(np.arange(64), np.arange(64), indexing="ij")
for th, tw in ((1, 64), (2, 32), (4, 16), (8, 8), (16, 4)):
    per_level = []
    box_level = []
    for l in range(4):
        f = np.floor(scene / 2 ** l).astype(np.int64)
        ox = f[..., 0] - 3 - (xx >> l)[None]
        oy = f[..., 1] - 3 - (yy >> l)[None]
        g = lambda a: a.reshape(N, 64 // th, th, 64 // tw, tw).transpose(0, 1, 3, 2, 4).reshape(N * (64 // th) * (64 // tw), 64)
        gx, gy = g(ox), g(oy)
        # pixels whose window leaves the level's map entirely read nothing; ignore (rare)
        cnt = 0
        box = 0
        for k in range(gx.shape[0]):
            x0, y0 = gx[k].min(), gy[k].min()
            nx, ny = gx[k].max() - x0 + 8, gy[k].max() - y0 + 8
            box += nx * ny
            if nx * ny > 4096:
                cnt += 64 * 64
                continue
            m = np.zeros((ny, nx), bool)
            for px, py in set(zip((gx[k] - x0).tolist(), (gy[k] - y0).tolist())):
                m[py:py + 8, px:px + 8] = True
            cnt += m.sum()
        per_level.append(cnt / gx.shape[0])
        box_level.append(box / gx.shape[0])
    print("%2d x %2d pixels per line: lines per group, levels 0-3: %s  mean %.1f   (bounding boxes: %s mean %.1f)" % (
        th, tw, " ".join("%6.1f" % v for v in per_level), np.mean(per_level), " ".join("%6.1f" % v for v in box_level), np.mean(box_level)))
