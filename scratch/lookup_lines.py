"""scratch: the streaming lookup's time against the number of 128-byte lines its strips touch (the cost is per
line).  96 edges, 64x64 maps, pyramids rotated so that the windows come from HBM; coordinate fields of growing roughness:
identity + constant shift (64 lines per strip and level: the algorithmic minimum), the bench scene, the bench scene + noise."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dba-fusion_amd"))
import numpy as np, torch
from dbaf_amd import synthetic as syn
from dbaf_amd.corr import CorrBlock
W = syn.window_25_96(0)
dev = "cuda"
fm = torch.from_numpy(syn.make_fmaps(W.B, 128, 64, 64, 1000)).to(dev)
ii, jj = torch.from_numpy(W.ii).to(dev), torch.from_numpy(W.jj).to(dev)
blocks = [CorrBlock(fm[ii][None], fm[jj][None]).build() for _ in range(3)]
yy, xx = np.meshgrid(np.arange(64, dtype=np.float32), np.arange(64, dtype=np.float32), indexing="ij")
ident = np.stack([xx, yy], -1)[None].repeat(W.N, 0)
scene, _ = syn.reproject_np(W.poses, W.disps, W.intrinsics, W.ii, W.jj)
rng = np.random.default_rng(0)


def lines(c):
    """mean number of distinct (dy, dx) offsets per 64-pixel strip and level = 128-byte lines read"""
    tot = 0.0
    for l in range(4):
        f = np.floor(c / 2 ** l).astype(np.int64)
        ox = f[..., 0] - (np.arange(64)[None, None, :] >> l)
        oy = f[..., 1] - (np.arange(64)[None, :, None] >> l)
        nx = ox.max(-1) - ox.min(-1) + 8
        ny = oy.max(-1) - oy.min(-1) + 8
        tot += (nx * ny).mean()
    return tot / 4


def tiled(c, th, tw):
    """the scene's FLOW re-arranged so that a 64 x 1 strip carries the flow of a th x tw pixel tile: the lines the kernel then
    reads are the lines a layout with th x tw tiles per 128-byte line would read on the real scene"""
    flow = c - ident
    N = c.shape[0]
    f = flow.reshape(N, 64 // th, th, 64 // tw, tw, 2).transpose(0, 1, 3, 2, 4, 5).reshape(N, 64, 64, 2)   # tile-major pixels
    return (ident + f).astype(np.float32)


for name, c in (("bench scene as 4 x 16 tiles", tiled(scene, 4, 16)), ("bench scene as 8 x 8 tiles", tiled(scene, 8, 8)),
                ("bench scene as 2 x 32 tiles", tiled(scene, 2, 32)),
                ("identity + (2.3, 1.7)", ident + np.array([2.3, 1.7], np.float32)),
                ("bench scene", scene.astype(np.float32)),
                ("bench scene + U(-1, 1) px", (scene + rng.uniform(-1, 1, scene.shape)).astype(np.float32)),
                ("bench scene + U(-2, 2) px", (scene + rng.uniform(-2, 2, scene.shape)).astype(np.float32))):
    ct = torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).to(dev)[None]
    keep = []
    for k in range(3):
        keep.append(blocks[k](ct))
    torch.cuda.synchronize()
    ts = []
    for k in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record()
        keep[k % 3] = blocks[k % 3](ct, timing=(a, b))
        ts.append((a, b))
    torch.cuda.synchronize()
    us = np.mean([a.elapsed_time(b) for a, b in ts]) * 1e3
    ln = lines(c)
    print("%-28s union box %5.1f lines read per strip and level (+ 49 written) -> %6.1f us = %.3f us per line read, %.3f of 8 TB/s on "
          "algorithmic bytes" % (name, ln, us, us / ln, 358.6e6 / (us * 1e-6) / 8e12), flush=True)
