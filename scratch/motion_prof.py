"""scratch: what the motion filter's unit (a one-edge CorrBlock + one lookup, dbaf/motion_filter.py:74-76) consists of"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
import torch
from dbaf_amd.corr import CorrBlock
torch.manual_seed(0)
h = w = 64
fm1 = torch.randn(1, 1, 128, h, w, device="cuda", dtype=torch.half) * 0.1
fm2 = torch.randn(1, 1, 128, h, w, device="cuda", dtype=torch.half) * 0.1
ys, xs = torch.meshgrid(torch.arange(h, device="cuda").float(), torch.arange(w, device="cuda").float(), indexing="ij")
coords = torch.stack([xs, ys], -1)[None, None] + 0.3
for _ in range(5):
    out = CorrBlock(fm1, fm2)(coords)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
a.record()
for _ in range(50):
    out = CorrBlock(fm1, fm2)(coords)
b.record()
torch.cuda.synchronize()
print("motion filter unit: %.1f us device, %.1f us wall per frame" % (a.elapsed_time(b) * 1e3 / 50, (time.perf_counter() - t0) * 1e6 / 50))
