"""scratch: only the matrix-core AltCorrBlock lookup on the bench window (for rocprofv3 passes)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
from dbaf_amd import synthetic as syn
from dbaf_amd import projective_ops as pops
from dbaf_amd.corr import AltCorrBlock
W = syn.window_25_96(0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
ii, jj = t(W.ii), t(W.jj)
K = t(W.intrinsics)[None, None].expand(1, W.B, 4).contiguous()
coords, _ = pops.projective_transform(t(W.poses)[None], t(W.disps)[None], K, ii, jj)
blk = AltCorrBlock(t(syn.make_fmaps(W.B, 128, W.h, W.w, 1000))[None], num_levels=4, radius=3)
with torch.no_grad():
    for _ in range(6):
        blk(coords, ii, jj)
torch.cuda.synchronize()
