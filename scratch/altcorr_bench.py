"""AltCorrBlock (on-the-fly correlation, no volume) on the bench window: 96 edges, 64x64 maps, 128 channels, 4 levels.
python scratch/altcorr_bench.py   (prints us per call and per edge-level for float and half maps)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
from dbaf_amd import synthetic as syn  # noqa: E402
from dbaf_amd import projective_ops as pops  # noqa: E402
import droid_backends  # noqa: E402

W = syn.window_25_96(0)
dev = "cuda"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
fm = t(syn.make_fmaps(W.B, 128, W.h, W.w, 1000)).float() / 4.0
ii, jj = t(W.ii), t(W.jj)
K = t(W.intrinsics)[None, None].expand(1, W.B, 4).contiguous()
coords, _ = pops.projective_transform(t(W.poses)[None], t(W.disps)[None], K, ii, jj)   # [1, N, h, w, 2]
pyr = []
f = fm
for lvl in range(4):
    pyr.append(f.permute(0, 2, 3, 1).contiguous())
    f = torch.nn.functional.avg_pool2d(f, 2, stride=2)
for dt in (torch.float32, torch.float16):
    f1 = pyr[0][ii].to(dt).contiguous()
    tot = 0.0
    for lvl in range(4):
        f2 = pyr[lvl][jj].to(dt).contiguous()
        c = (coords[0] / 2 ** lvl)[:, None].contiguous()
        for _ in range(2):
            droid_backends.altcorr_forward(f1, f2, c, 3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            droid_backends.altcorr_forward(f1, f2, c, 3)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        tot += us
        flops = 2.0 * W.N * W.h * W.w * 64 * 128
        print("%s level %d: %.1f us for %d edges (%.2f us/edge, %.1f TFLOP/s of window dot products)" % (
            str(dt).split(".")[1], lvl, us, W.N, us / W.N, flops / us / 1e6), flush=True)
    print("%s 4 levels: %.1f us (the volume lookup of the same window: ~95 us + 24 us/edge once for the volume)" % (
        str(dt).split(".")[1], tot), flush=True)

# the whole AltCorrBlock lookup as the caller issues it (round 4: one launch for the four levels, maps indexed by ii / jj in
# the kernel) against the per-level route (gathers + scaled coordinate copies + four launches + slice copies)
from dbaf_amd.corr import AltCorrBlock  # noqa: E402
blk = AltCorrBlock(t(syn.make_fmaps(W.B, 128, W.h, W.w, 1000)).float()[None], num_levels=4, radius=3)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


with torch.no_grad():
    fused = timed(lambda: blk(coords, ii, jj))
cg = coords.clone().requires_grad_(True)
per_level = timed(lambda: blk(cg, ii, jj))
print("AltCorrBlock.__call__, 96 edges, float: one launch %.1f us; per-level route (autograd) %.1f us" % (fused, per_level), flush=True)

# the reference's case: HALF maps (autocast) -> the matrix-core form (one small GEMM per 4 x 16 tile and level)
blk_h = AltCorrBlock(t(syn.make_fmaps(W.B, 128, W.h, W.w, 1000))[None], num_levels=4, radius=3)
assert blk_h.pyramid[0].dtype == torch.float16
with torch.no_grad():
    mf = timed(lambda: blk_h(coords, ii, jj))
    a = blk_h(coords, ii, jj)
    blk_h.mfma = False
    fl = timed(lambda: blk_h(coords, ii, jj))
    b = blk_h(coords, ii, jj)
print("AltCorrBlock.__call__ on HALF maps, 96 edges: matrix cores %.1f us, float chain on the .float() twins %.1f us; max |diff| %.2e "
      "(|corr| max %.2f)" % (mf, fl, float((a - b).abs().max()), float(b.abs().max())), flush=True)
