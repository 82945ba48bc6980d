"""Times the fused volume build (CorrBlock(fmap1, fmap2)) per edge for the config map shapes and checks it bit for bit
against the unfused pipeline on two edges.   python scratch/build_ab.py [tag]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
from dbaf_amd import synthetic as syn  # noqa: E402
from dbaf_amd.corr import CorrBlock  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
for (h, w) in ((64, 64), (28, 107), (55, 55), (48, 64)):
    fm = torch.from_numpy(syn.make_fmaps(33, 128, h, w, 1)).cuda()
    f1, f2 = fm[:32][None], fm[1:33][None]
    a = CorrBlock.build_sheared_fused(f1[:, :2], f2[:, :2], 4)
    b = CorrBlock.shear_pyramid(CorrBlock.build_pyramid(f1[:, :2], f2[:, :2], 4))
    ok = all(torch.equal(CorrBlock.map_pixels(x, h, w).contiguous().view(torch.int16), CorrBlock.map_pixels(y, h, w).contiguous().view(torch.int16))
             for x, y in zip(a, b))
    del a, b
    for _ in range(2):
        CorrBlock.build_sheared_fused(f1, f2, 4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        CorrBlock.build_sheared_fused(f1, f2, 4)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5 / 32
    hw = h * w
    by = 2 * 128 * hw * 2 + int(hw * hw * (1 + .25 + .0625 + .015625)) * 2
    print("%-10s %dx%d  %.2f us/edge  %.2f TB/s (%.3f of 8)  %.0f TFLOP/s  bit-exact=%s" % (
        tag, h, w, us, by / us / 1e6, by / us / 1e6 / 8, 2 * hw * hw * 128 / us / 1e6, ok), flush=True)
