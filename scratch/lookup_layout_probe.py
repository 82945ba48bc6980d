"""What would tiled planes buy the ragged map shapes?  The same window (graph, poses, depth statistics, intrinsics scaled with the width)
at the real width (linear planes: the resident lookup) and at the next width whose planes are tiled (a multiple of 64: the
rows-over-tiles lookup), lookups with the reprojection in the launch, rotating over three pyramid copies like bench.py.
Reported per PIXEL of the map, so that the two widths compare:   python scratch/lookup_layout_probe.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
from dbaf_amd import synthetic as syn  # noqa: E402
from dbaf_amd.corr import CorrBlock  # noqa: E402

dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
for name, graph, nkf, h, widths, intr in (("KITTI 32/122", syn.graph_32_122(), 32, 28, (107, 128), syn.TUMVI_INTRINSICS_8),
                                          ("TUM-VI 9/36", syn.graph_banded(9, 2, extra=[(0, 3), (1, 4), (2, 5)]), 9, 55, (55, 64), (20.5, 20.5, 27.4, 27.6))):
    for w in widths:
        hh = h if w != 64 else 56
        sc = w / widths[0]
        k = (intr[0] * sc, intr[1] * (hh / h), intr[2] * sc, intr[3] * (hh / h))
        W = syn.make_window(*graph, nkf, hh, w, seed=3, intr=k)
        poses, disps, K = t(W.poses), t(W.disps), t(W.intrinsics)[None].expand(W.B, 4).contiguous()
        ii, jj = t(W.ii), t(W.jj)
        fm = t(syn.make_fmaps(W.B, 128, hh, w, 5))
        blocks = [CorrBlock(fm[ii][None], fm[jj][None]).build() for _ in range(3)]
        for i in range(6):
            blocks[i % 3].lookup_reprojected(poses, disps, K, ii, jj)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30):
            out = blocks[i % 3].lookup_reprojected(poses, disps, K, ii, jj)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        npx = W.N * hh * w
        alg = npx * (512 + 8 + 392)
        print("%-13s %3dx%-3d %-8s %7.1f us per lookup  %.3f ns per pixel and edge  %.3f of the HBM peak (algorithmic bytes)" % (
            name, hh, w, "tiled" if (w % 64 == 0 and hh % 4 == 0) else "linear", us, us * 1e3 / npx, alg / us / 1e6 / 8000), flush=True)
        del blocks, fm
        torch.cuda.empty_cache()
