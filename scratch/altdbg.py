import sys, numpy as np, torch
sys.path[:0] = ['/root/repo', '/root/repo/dba-fusion_amd', '/root/repo/tests']
import droid_backends
from oracle import oracle as orc
import test_gpu_corr as T
rng = np.random.default_rng(9)
B, S, H1, W1, H2, W2, C, r = 2, 1, 24, 40, 24, 40, 128, 3
f1, f2, coords = T._alt_case(rng, B, S, H1, W1, H2, W2, C, True)
f1, f2 = f1.astype(np.float16), f2.astype(np.float16)
ref = orc.altcorr_forward(f1, f2, coords, r)
out, = droid_backends.altcorr_forward(torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda(), torch.from_numpy(coords).cuda(), r)
got = out.cpu().numpy()
h = np.float16
def emulate(b, h1, w1, chunk, fma, order="k"):
    x2, y2 = coords[b, 0, h1, w1]
    fx, fy = np.floor(x2), np.floor(y2)
    dx, dy = np.float32(x2 - fx), np.float32(y2 - fy)
    wnw, wne = h(dy * dx), h(dy * (np.float32(1) - dx))
    wsw, wse = h((np.float32(1) - dy) * dx), h((np.float32(1) - dy) * (np.float32(1) - dx))
    acc = np.zeros((7, 7), h)  # [iy][ix]
    for c0 in range(0, C, chunk):
        for iy in range(8):
            for ix in range(8):
                h2, w2 = int(fy) - 3 + iy, int(fx) - 3 + ix
                s = h(0)
                if 0 <= h2 < H2 and 0 <= w2 < W2:
                    for k in range(chunk):
                        a, bb = f1[b, h1, w1, c0 + k], f2[b, h2, w2, c0 + k]
                        if fma:
                            s = h(np.float64(s) + np.float64(a) * np.float64(bb))
                        else:
                            s = h(np.float32(s) + np.float32(h(np.float32(a) * np.float32(bb))))
                if iy > 0 and ix > 0: acc[iy-1, ix-1] = h(np.float32(acc[iy-1, ix-1]) + np.float32(h(np.float32(s) * np.float32(wnw))))
                if iy > 0 and ix < 7: acc[iy-1, ix] = h(np.float32(acc[iy-1, ix]) + np.float32(h(np.float32(s) * np.float32(wne))))
                if iy < 7 and ix > 0: acc[iy, ix-1] = h(np.float32(acc[iy, ix-1]) + np.float32(h(np.float32(s) * np.float32(wsw))))
                if iy < 7 and ix < 7: acc[iy, ix] = h(np.float32(acc[iy, ix]) + np.float32(h(np.float32(s) * np.float32(wse))))
    return acc.T.reshape(49)  # channel = iy + 7 * ix
g = got[0, 0, :, 13, 3]; rf = ref[0, 0, :, 13, 3]
for name, kw in (("oracle order", dict(chunk=32, fma=False)), ("fma dot", dict(chunk=32, fma=True)), ("chunk16", dict(chunk=16, fma=False))):
    e = emulate(0, 13, 3, **kw)
    print(name, "== ref:", int((e.view(np.uint16) == rf.view(np.uint16)).sum()), "== got:", int((e.view(np.uint16) == g.view(np.uint16)).sum()), "of 49")
print("f1 pixel finite:", np.isfinite(f1[0, 13, 3].astype(np.float32)).all(), "max|f1|", np.abs(f1[0,13,3].astype(np.float32)).max())
