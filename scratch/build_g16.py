"""Times the fused volume build per edge on maps whose planes keep the linear pixel order (32 < w <= 64: the general sixteen-wave
walk, corr_build_fused16g_kernel; DBA_BUILD_WAVES=8 keeps the eight-wave walk) and checks it bit for bit against the unfused
pipeline on two edges.   python scratch/build_g16.py [tag] [n_edges]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
from dbaf_amd import synthetic as syn  # noqa: E402
from dbaf_amd.corr import CorrBlock  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
for (h, w) in [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SHAPES", "55x55,40x56,30x40,44x60,64x64").split(",")]:
    fm = torch.from_numpy(syn.make_fmaps(n + 1, 128, h, w, 1)).cuda()
    f1, f2 = fm[:n][None], fm[1:n + 1][None]
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
    us = e0.elapsed_time(e1) * 1e3 / 5 / n
    hw = h * w
    by = 2 * 128 * hw * 2 + sum((h >> l) * (w >> l) for l in range(4)) * hw * 2
    print("%-10s %dx%d n=%d  %.2f us/edge  %.2f TB/s (%.3f of 8)  bit-exact=%s" % (tag, h, w, n, us, by / us / 1e6, by / us / 1e6 / 8, ok), flush=True)
