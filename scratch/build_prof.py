import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
from dbaf_amd import synthetic as syn
from dbaf_amd.corr import CorrBlock
h, w = 64, 64
fm = torch.from_numpy(syn.make_fmaps(33, 128, h, w, 1)).cuda()
for n in (32,):
    f1, f2 = fm[:n][None], fm[1:n + 1][None]
    for _ in range(2):
        CorrBlock.build_sheared_fused(f1, f2, 4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    CorrBlock.build_sheared_fused(f1, f2, 4)
    e1.record()
    torch.cuda.synchronize()
    print("n=%d  %.2f us/edge (incl. profile read-back)" % (n, e0.elapsed_time(e1) * 1e3 / n))
