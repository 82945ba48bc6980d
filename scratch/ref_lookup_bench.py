"""scratch: the reference-layout lookup (droid_backends.corr_index_forward, shadows off) on the 96-edge 64x64 window, from HBM
(rotating pyramid copies).  DBA_HIP_LIB selects a variant build."""
import os, sys, time
os.environ["DBA_ZERO_EDIT_SHADOW"] = "0"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dba-fusion_amd"))
import numpy as np, torch
from dbaf_amd import synthetic as syn
from dbaf_amd.corr import CorrBlock
import droid_backends
W = syn.window_25_96(0)
dev = "cuda"
fm = torch.from_numpy(syn.make_fmaps(W.B, 128, 64, 64, 1000)).to(dev)
ii, jj = torch.from_numpy(W.ii).to(dev), torch.from_numpy(W.jj).to(dev)
pyrs = [CorrBlock.build_pyramid(fm[ii][None], fm[jj][None], 4) for _ in range(3)]
c = torch.from_numpy(syn.lookup_coords(W)).to(dev).permute(0, 3, 1, 2).contiguous()
cs = [(c / 2 ** l).contiguous() for l in range(4)]
def run(k):
    return [droid_backends.corr_index_forward(pyrs[k % 3][l], cs[l], 3)[0] for l in range(4)]
for k in range(3): keep = run(k)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
n = 12
for k in range(n): keep = run(k)
b.record(); torch.cuda.synchronize()
print("%s: 4-level reference-layout lookup, 96 edges: %.1f us" % (os.environ.get("DBA_HIP_LIB", "default").split("/")[-1], a.elapsed_time(b) * 1e3 / n))
