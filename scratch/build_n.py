"""Times the fused volume build for n edges per call (the batches add_factors / the motion filter really build) at 64x64 and 48x64.
   python scratch/build_n.py [tag]      (DBA_BUILD_WG_TARGET=256: the launch rule before the round's last session)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
from dbaf_amd import synthetic as syn  # noqa: E402
from dbaf_amd.corr import CorrBlock  # noqa: E402
tag = sys.argv[1] if len(sys.argv) > 1 else "default"
for (h, w) in ((64, 64), (48, 64), (55, 55)):
    fm = torch.from_numpy(syn.make_fmaps(97, 128, h, w, 1)).cuda()
    line = []
    for n in [int(x) for x in os.environ.get("N_LIST", "1,2,6,12,32,54,96").split(",")]:
        f1, f2 = fm[:n][None], fm[1:n + 1][None]
        for _ in range(3):
            CorrBlock.build_sheared_fused(f1, f2, 4)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if n <= 12 else 5
        e0.record()
        for _ in range(reps):
            CorrBlock.build_sheared_fused(f1, f2, 4)
        e1.record()
        torch.cuda.synchronize()
        line.append("n=%d: %.1f us (%.2f/edge)" % (n, e0.elapsed_time(e1) * 1e3 / reps, e0.elapsed_time(e1) * 1e3 / reps / n))
    print("%-8s %dx%d  " % (tag, h, w) + "  ".join(line), flush=True)
