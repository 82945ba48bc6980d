import os, sys, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
import numpy as np, torch
if len(sys.argv) > 1:
    from dbaf_amd.corr import CorrBlock
    n, C, h, w = 1, 128, int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(12)
    t1 = torch.from_numpy(rng.standard_normal((1, n, C, h, w)).astype(np.float16)).cuda()
    t2 = torch.from_numpy(rng.standard_normal((1, n, C, h, w)).astype(np.float16)).cuda()
    f = CorrBlock.build_sheared_fused(t1, t2, 4)
    np.save(sys.argv[1], CorrBlock.map_pixels(f[0], h, w).cpu().numpy().view(np.uint16))
else:
    for h, w in ((64, 64), (24, 40)):
        for o in ("copy", "native"):
            subprocess.check_call([sys.executable, __file__, "/tmp/nd_%s.npy" % o, str(h), str(w)], env=dict(os.environ, DBA_BUILD_OPERANDS=o), timeout=120)
        a, b = np.load("/tmp/nd_copy.npy"), np.load("/tmp/nd_native.npy")
        print(h, w, a.shape, "mismatch fraction", (a != b).mean())
        bad = np.argwhere(a != b)
        if len(bad):
            print(" first bad", bad[:5].tolist(), " bad per axis uniques:", [len(np.unique(bad[:, k])) for k in range(bad.shape[1])], "of", a.shape)
            af, bf = a.view(np.float16).astype(np.float32), b.view(np.float16).astype(np.float32)
            print(" max abs diff", np.abs(af - bf).max(), "mean abs", np.abs(af).mean())
