"""scratch: does the lookup's time depend on where the allocator puts the pyramids / outputs?  Same box, one process per
offset: a dummy allocation of varying size first, then the bench window's lookups from HBM."""
import os, sys, subprocess
if len(sys.argv) > 1:
    import numpy as np, torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
    from dbaf_amd import synthetic as syn
    from dbaf_amd.corr import CorrBlock
    from dbaf_amd import projective_ops as pops
    off = int(sys.argv[1])
    dummy = torch.empty(max(off, 1), dtype=torch.uint8, device="cuda")
    W = syn.window_25_96(0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    fm = t(syn.make_fmaps(W.B, 128, 64, 64, 1000)); ii, jj = t(W.ii), t(W.jj)
    blocks = [CorrBlock(fm[ii][None], fm[jj][None]).build() for _ in range(3)]
    K = t(W.intrinsics)[None, None].expand(1, W.B, 4).contiguous()
    coords, _ = pops.projective_transform(t(W.poses)[None], t(W.disps)[None], K, ii, jj)
    keep = [b(coords) for b in blocks]
    torch.cuda.synchronize()
    ts = []
    for k in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record()
        keep[k % 3] = blocks[k % 3](coords, timing=(a, b))
        ts.append((a, b))
    torch.cuda.synchronize()
    us = sorted(x.elapsed_time(y) * 1e3 for x, y in ts[6:])
    print("offset %10d B: lookup p10 %.1f p50 %.1f p90 %.1f us; store0 ptr %% 2MiB = %d KiB, out ptr %% 2MiB = %d KiB" % (
        off, us[2], us[len(us) // 2], us[-3], (blocks[0]._stores[0].data_ptr() % (2 << 20)) >> 10, (keep[0].data_ptr() % (2 << 20)) >> 10), flush=True)
else:
    for off in (0, 4096, 65536, 1 << 20, 3 << 20, 17 << 20, (64 << 20) + 8192, 200 << 20, 1 << 30):
        subprocess.run([sys.executable, os.path.abspath(__file__), str(off)], stderr=subprocess.DEVNULL)
