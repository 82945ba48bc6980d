import sys, numpy as np, torch
sys.path.insert(0,'dba-fusion_amd'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from dbaf_amd.corr import CorrBlock
from oracle import oracle as orc
import droid_backends
from test_gpu_corr import _coords
rng = np.random.default_rng(5)
n, C, h, w = 3, 32, 16, 24
f1 = rng.standard_normal((1, n, C, h, w)).astype(np.float16)
f2 = rng.standard_normal((1, n, C, h, w)).astype(np.float16)
cb = CorrBlock(torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda(), num_levels=4, radius=3)
pyr_gpu = [p.cpu().numpy() for p in cb.corr_pyramid]
coords = _coords(rng, n, h, w, h, w).transpose(0, 2, 3, 1)
out = cb(torch.from_numpy(np.ascontiguousarray(coords))[None].cuda()).cpu().numpy()[0]
ref = orc.corr_lookup_pyramid(pyr_gpu, coords, 3)
for l in range(4):
    a = out[:, l*49:(l+1)*49].astype(np.float32); b = ref[:, l*49:(l+1)*49].astype(np.float32)
    bad = (a != b)
    print('level', l, 'mismatch frac', bad.mean(), 'max diff', np.abs(a-b).max())
    if bad.any():
        idx = np.argwhere(bad)[:5]
        for i in idx: print('  ', i, a[tuple(i)], b[tuple(i)], 'coords', coords[i[0], i[2], i[3]])
# f32 single level
vol = (rng.standard_normal((2, 16, 16, 16, 16)) * 4).astype(np.float32)
c2 = _coords(rng, 2, 16, 16, 16, 16)
r = orc.corr_index_forward(vol, c2, 3)
o, = droid_backends.corr_index_forward(torch.from_numpy(vol).cuda(), torch.from_numpy(c2).cuda(), 3)
o = o.cpu().numpy()
bad = o != r
print('f32 mismatch frac', bad.mean(), 'max abs', np.abs(o-r).max(), 'max rel', (np.abs(o-r)/np.maximum(np.abs(r),1e-9))[bad].max() if bad.any() else 0)
