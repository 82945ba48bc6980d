import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
from dbaf_amd import synthetic as syn
from dbaf_amd.corr import CorrBlock
h, w = 64, 64
fm = torch.from_numpy(syn.make_fmaps(33, 128, h, w, 1)).cuda()
for n in (1, 8, 32):
    f1, f2 = fm[:n][None], fm[1:n + 1][None]
    for _ in range(3):
        CorrBlock.build_sheared_fused(f1, f2, 4)
