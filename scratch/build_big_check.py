"""bit-for-bit check of the fused volume build against the unfused pipeline at large edge counts (more workgroups than one round, every
walk length the launch rule picks): python scratch/build_big_check.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
from dbaf_amd.corr import CorrBlock  # noqa: E402

for (n, h, w) in ((96, 55, 55), (37, 55, 55), (64, 28, 107), (13, 28, 107), (50, 44, 60), (20, 61, 61), (70, 33, 65), (96, 64, 64)):
    rng = np.random.default_rng(n + h)
    t1 = torch.from_numpy(rng.standard_normal((1, n, 128, h, w)).astype(np.float16)).cuda()
    t2 = torch.from_numpy(rng.standard_normal((1, n, 128, h, w)).astype(np.float16)).cuda()
    fused = CorrBlock.build_sheared_fused(t1, t2, 4)
    ok = True
    for c0 in range(0, n, 16):   # (the unfused pipeline in chunks: its reference-layout levels are twice the memory)
        unf = CorrBlock.shear_pyramid(CorrBlock.build_pyramid(t1[:, c0:c0 + 16], t2[:, c0:c0 + 16], 4))
        for lvl in range(4):
            a = CorrBlock.map_pixels(fused[lvl][c0:c0 + 16], h, w).contiguous().view(torch.int16)
            b = CorrBlock.map_pixels(unf[lvl], h, w).contiguous().view(torch.int16)
            ok = ok and bool(torch.equal(a, b))
        del unf
    print("n=%-3d %dx%d  fused == unfused: %s" % (n, h, w, ok), flush=True)
    del fused, t1, t2
    torch.cuda.empty_cache()
