import sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, 'dba-fusion_amd'); sys.path.insert(0, '.')
import test_gpu_corr as T
from dbaf_amd.corr import CorrBlock
orc = T._oracle()
rng = np.random.default_rng(5)
n, C, h, w = (2, 16, 24, 64)
f1 = rng.standard_normal((1, n, C, h, w)).astype(np.float16); f2 = rng.standard_normal((1, n, C, h, w)).astype(np.float16)
t1, t2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
cb = CorrBlock(t1, t2, num_levels=4, radius=3, layout="sheared")
pyr_ref = [p.cpu().numpy() for p in CorrBlock.build_pyramid(t1, t2, 4)]
for name, coords in (("random", T._coords(rng, n, h, w, h, w).transpose(0, 2, 3, 1)), ("smooth", T._smooth_coords(rng, n, h, w))):
    coords = np.ascontiguousarray(coords)
    out = cb(torch.from_numpy(coords)[None].cuda()).cpu().numpy()[0]
    ref = orc.corr_lookup_pyramid(pyr_ref, coords, 3)
    bad = out.view(np.uint16) != ref.view(np.uint16)
    print(name, "mismatch frac", bad.mean())
    if bad.any():
        idx = np.argwhere(bad)
        print(" per level:", [bad[:, 49*l:49*(l+1)].mean() for l in range(4)])
        print(" first:", idx[:5].tolist(), out[tuple(idx[0])], ref[tuple(idx[0])])
        e, ch, y, x = idx[0]; print(" coords", coords[e, y, x], " nan in out:", np.isnan(out.astype(np.float32)).sum())
        print(" rows with mismatch:", np.unique(idx[:,2])[:30], " x:", np.unique(idx[:,3])[:70])
