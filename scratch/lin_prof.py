# experiment: phase timestamps of the linearisation kernel (ablation build with -DLIN_PROF), one ba iteration
import sys, ctypes, numpy as np, torch
sys.path.insert(0, 'dba-fusion_amd')
from dbaf_amd import synthetic as syn, _lib
import droid_backends
lib = _lib.load()
W = syn.window_25_96(0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
N, B, ht, wd = len(W.ii), W.B, 64, 64
dims = (N, B, ht, wd, W.t0, W.t1)
nbytes = lib.dba_ba_workspace_bytes(*dims)
lay = _lib.BaLayout(); lib.dba_ba_get_layout(*dims, ctypes.byref(lay))
ws = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')
poses, disps = t(W.poses), t(W.disps)
intr = t(W.intrinsics); dsens = t(W.disps_sens); target, weight, eta = t(W.target), t(W.weight), t(W.eta)
ii, jj = t(W.ii), t(W.jj)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda x: ctypes.c_void_p(x.data_ptr())
lib.dba_ba_prepare(P(ii), P(jj), N, B, ht, wd, W.t0, W.t1, P(ws), nbytes, st)
nblk = 16 * (min(B, N + W.t1 - W.t0) + 1)
prof = torch.zeros(nblk * 8, dtype=torch.int32, device='cuda')
meta = ws[lay.meta:lay.meta + 128].view(torch.int64)
meta[8] = prof.data_ptr()
for rep in range(4):
    prof.zero_()
    rc = lib.dba_ba_linearize(P(poses), P(disps), P(intr), P(dsens), P(target), P(weight), P(eta), eta.shape[0] if eta.dim() > 1 else 1,
                              P(ii), P(jj), None, N, B, ht, wd, W.t0, W.t1, ctypes.c_float(0.05), P(ws), nbytes, st)
    torch.cuda.synchronize()
    m = prof.cpu().numpy().reshape(-1, 8)
    m = m[m[:, 4] != 0]          # blocks that ran to the end
    t0 = m[:, 0].min()
    r = lambda c: (int(np.percentile(m[:, c] - t0, 50)), int((m[:, c] - t0).max()))
    print('rc', rc, 'blocks', len(m), 'ticks(10ns) from first block start, median/max: start', r(0), 'poses staged', r(2), 'edge loop done', r(3), 'end', r(4), '| cycles per block in: pixel+E', int(np.median(m[:,5])), 'staging', int(np.median(m[:,6])), 'mfma+scatter', int(np.median(m[:,7])))
