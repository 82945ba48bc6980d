"""fused volume build: us per edge for a batch of 32 edges (CorrBlock(f1, f2).build(), and the same edges built into free slots
of a standing block), under DBA_BUILD_OPERANDS=copy|native.  python scratch/build_bench.py [h w]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
import torch
from dbaf_amd.corr import CorrBlock

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 64)
C, nb = 128, 32
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(1)
fm = torch.randn(40, C, h, w, generator=g).half().to(dev)
ii = torch.randint(0, 40, (64,), generator=g).to(dev)
jj = torch.randint(0, 40, (64,), generator=g).to(dev)
f1, f2 = fm[ii[:nb]][None], fm[jj[:nb]][None]


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


b = timed(lambda: CorrBlock(f1, f2, num_levels=4, radius=3).build(), 20) / nb
host = CorrBlock(fm[ii][None], fm[jj][None], num_levels=4, radius=3, capacity=64).build()
keep = torch.arange(64, device=dev) >= 32
st = {"blk": host}


def into():
    st["blk"] = st["blk"][keep].cat(CorrBlock(f1, f2, num_levels=4, radius=3))


s = timed(into, 20) / nb
print("%dx%d operands=%s: build %.2f us/edge, into slots %.2f us/edge" % (h, w, os.environ.get("DBA_BUILD_OPERANDS", "native"), b, s))
