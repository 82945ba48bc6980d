"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the CPU oracle (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (dba-fusion_amd/) must never import it.

All functions take/return numpy arrays.  `dtype` selects the instantiation:
np.float32 = faithful fp32 restatement, np.float64 = fp64 arbiter.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ba_oracle.c", "corr_oracle.c", "ba_impl.inc", "half.h")]
    stale = (not os.path.exists(so)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _sfx(dtype):
    return {np.dtype(np.float32): "_f32", np.dtype(np.float64): "_f64"}[np.dtype(dtype)]


def _real(dtype):
    return ctypes.c_float if np.dtype(dtype) == np.float32 else ctypes.c_double


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def ba(poses, disps, intr, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations=2,
       lm=1e-4, ep=0.1, motion_only=False, alpha=0.05, dtype=np.float32):
    """droid_backends.ba (droid_kernels.cu:1394-1512). Returns dict with updated copies."""
    L = lib()
    poses = _c(poses, dtype).copy()
    disps = _c(disps, dtype).copy()
    B, ht, wd = disps.shape
    intr = _c(intr, dtype)
    disps_sens = _c(disps_sens, dtype)
    targets = _c(targets, dtype)
    weights = _c(weights, dtype)
    eta = _c(eta, dtype).reshape(-1, ht * wd)
    ii = _c(ii, np.int64)
    jj = _c(jj, np.int64)
    N = ii.shape[0]
    P = t1 - t0
    dx = np.zeros((max(P, 0), 6), dtype)
    dz = np.zeros((P + N, ht * wd), dtype)
    M = ctypes.c_int(0)
    R = _real(dtype)
    fn = getattr(L, "oracle_ba" + _sfx(dtype))
    fn.restype = ctypes.c_int
    ok = fn(_p(poses), _p(disps), _p(intr), _p(disps_sens), _p(targets), _p(weights), _p(eta),
            ctypes.c_int(eta.shape[0]), _p(ii), _p(jj), ctypes.c_int(N), ctypes.c_int(B),
            ctypes.c_int(ht), ctypes.c_int(wd), ctypes.c_int(t0), ctypes.c_int(t1),
            ctypes.c_int(iterations), R(lm), R(ep), ctypes.c_int(int(motion_only)), R(alpha),
            _p(dx), _p(dz), ctypes.byref(M))
    return dict(poses=poses, disps=disps, dx=dx, dz=dz[:M.value], M=M.value, ok=bool(ok))


class BACore:
    """droid_backends.BACore (droid_kernels.cu:1786-1956); mutates its own copies."""

    def __init__(self, poses, disps, intr, disps_sens, targets, weights, eta, ii, jj, t0, t1,
                 lm=1e-4, ep=0.1, dtype=np.float32):
        self.L = lib()
        self.dtype = np.dtype(dtype)
        self.poses = _c(poses, dtype).copy()
        self.disps = _c(disps, dtype).copy()
        B, ht, wd = self.disps.shape
        self.ht, self.wd, self.P, self.N = ht, wd, t1 - t0, len(ii)
        self._keep = [_c(intr, dtype), _c(disps_sens, dtype), _c(targets, dtype), _c(weights, dtype),
                      _c(eta, dtype).reshape(-1, ht * wd), _c(ii, np.int64), _c(jj, np.int64)]
        k = self._keep
        R = _real(dtype)
        fn = getattr(self.L, "oracle_bacore_create" + _sfx(dtype))
        fn.restype = ctypes.c_void_p
        self.h = ctypes.c_void_p(fn(_p(self.poses), _p(self.disps), _p(k[0]), _p(k[1]), _p(k[2]), _p(k[3]),
                                    _p(k[4]), ctypes.c_int(k[4].shape[0]), _p(k[5]), _p(k[6]),
                                    ctypes.c_int(self.N), ctypes.c_int(B), ctypes.c_int(ht),
                                    ctypes.c_int(wd), ctypes.c_int(t0), ctypes.c_int(t1), R(lm), R(ep)))
        fm = getattr(self.L, "oracle_bacore_M" + _sfx(dtype))
        fm.restype = ctypes.c_int
        self.M = fm(self.h)

    def hessian(self):
        n = 6 * self.P
        H = np.zeros((n, n), np.float64)
        v = np.zeros((n,), np.float64)
        getattr(self.L, "oracle_bacore_hessian" + _sfx(self.dtype))(self.h, _p(H), _p(v))
        return H, v

    def optimize(self, H, v):
        dx = np.zeros((self.P, 6), self.dtype)
        fn = getattr(self.L, "oracle_bacore_optimize" + _sfx(self.dtype))
        fn.restype = ctypes.c_int
        ok = fn(self.h, _p(_c(H, np.float64)), _p(_c(v, np.float64)), _p(dx))
        return dx, bool(ok)

    def retract(self, dx64):
        dx64 = _c(dx64, np.float64).reshape(-1)
        dx = np.zeros((self.P, 6), self.dtype)
        dz = np.zeros((self.M, self.ht * self.wd), self.dtype)
        getattr(self.L, "oracle_bacore_retract" + _sfx(self.dtype))(self.h, _p(dx64), _p(dx), _p(dz))
        return dx, dz

    def get_EQw(self):
        HW = self.ht * self.wd
        E = np.zeros((self.P + self.N, 6, HW), self.dtype)
        Q = np.zeros((self.M, HW), self.dtype)
        w = np.zeros((self.M, HW), self.dtype)
        getattr(self.L, "oracle_bacore_get_EQw" + _sfx(self.dtype))(self.h, _p(E), _p(Q), _p(w))
        return E, Q, w

    def __del__(self):
        try:
            if getattr(self, "h", None):
                getattr(self.L, "oracle_bacore_destroy" + _sfx(self.dtype))(self.h)
                self.h = None
        except Exception:
            pass


def linearize(poses, disps, intr, targets, weights, ii, jj, dtype=np.float32):
    """projective_transform_kernel (droid_kernels.cu:220-468) raw outputs."""
    L = lib()
    poses, disps, intr = _c(poses, dtype), _c(disps, dtype), _c(intr, dtype)
    targets, weights = _c(targets, dtype), _c(weights, dtype)
    ii, jj = _c(ii, np.int64), _c(jj, np.int64)
    N = len(ii)
    _, ht, wd = disps.shape
    HW = ht * wd
    Hs = np.zeros((4, N, 6, 6), dtype); vs = np.zeros((2, N, 6), dtype)
    Eii = np.zeros((N, 6, HW), dtype); Eij = np.zeros((N, 6, HW), dtype)
    Cii = np.zeros((N, HW), dtype); bz = np.zeros((N, HW), dtype)
    getattr(L, "oracle_linearize" + _sfx(dtype))(
        _p(poses), _p(disps), _p(intr), _p(targets), _p(weights), _p(ii), _p(jj), ctypes.c_int(N),
        ctypes.c_int(ht), ctypes.c_int(wd), _p(Hs), _p(vs), _p(Eii), _p(Eij), _p(Cii), _p(bz))
    return dict(Hs=Hs, vs=vs, Eii=Eii, Eij=Eij, Cii=Cii, bz=bz)


def pose_retr(poses, dx, t0, t1, dtype=np.float32):
    poses = _c(poses, dtype).copy()
    dx = _c(dx, dtype)
    getattr(lib(), "oracle_pose_retr" + _sfx(dtype))(_p(poses), _p(dx), ctypes.c_int(t0), ctypes.c_int(t1))
    return poses


def frame_distance(poses, disps, intr, ii, jj, beta, dtype=np.float32):
    poses, disps, intr = _c(poses, dtype), _c(disps, dtype), _c(intr, dtype)
    ii, jj = _c(ii, np.int64), _c(jj, np.int64)
    _, ht, wd = disps.shape
    dist = np.zeros((len(ii),), dtype)
    getattr(lib(), "oracle_frame_distance" + _sfx(dtype))(
        _p(poses), _p(disps), _p(intr), _p(ii), _p(jj), ctypes.c_int(len(ii)), ctypes.c_int(ht),
        ctypes.c_int(wd), _real(dtype)(beta), _p(dist))
    return dist


def projmap(poses, disps, intr, ii, jj, dtype=np.float32):
    poses, disps, intr = _c(poses, dtype), _c(disps, dtype), _c(intr, dtype)
    ii, jj = _c(ii, np.int64), _c(jj, np.int64)
    _, ht, wd = disps.shape
    coords = np.zeros((len(ii), ht, wd, 3), dtype)
    valid = np.zeros((len(ii), ht, wd, 1), dtype)
    getattr(lib(), "oracle_projmap" + _sfx(dtype))(
        _p(poses), _p(disps), _p(intr), _p(ii), _p(jj), ctypes.c_int(len(ii)), ctypes.c_int(ht),
        ctypes.c_int(wd), _p(coords), _p(valid))
    return coords, valid


def iproj(poses, disps, intr, dtype=np.float32):
    poses, disps, intr = _c(poses, dtype), _c(disps, dtype), _c(intr, dtype)
    nm, ht, wd = disps.shape
    pts = np.zeros((nm, ht, wd, 3), dtype)
    getattr(lib(), "oracle_iproj" + _sfx(dtype))(
        _p(poses), _p(disps), _p(intr), ctypes.c_int(nm), ctypes.c_int(ht), ctypes.c_int(wd), _p(pts))
    return pts


def depth_filter(poses, disps, intr, inds, thresh, dtype=np.float32):
    poses, disps, intr = _c(poses, dtype), _c(disps, dtype), _c(intr, dtype)
    inds = _c(inds, np.int64)
    thresh = _c(thresh, dtype)
    nbuf, ht, wd = disps.shape
    counter = np.zeros((len(inds), ht, wd), dtype)
    getattr(lib(), "oracle_depth_filter" + _sfx(dtype))(
        _p(poses), _p(disps), _p(intr), _p(inds), _p(thresh), ctypes.c_int(len(inds)),
        ctypes.c_int(nbuf), ctypes.c_int(ht), ctypes.c_int(wd), _p(counter))
    return counter


def reproject(poses, disps, intr_b, ii, jj, dtype=np.float32):
    """pops.projective_transform without jacobians (projective_ops.py:96-125)."""
    poses, disps = _c(poses, dtype), _c(disps, dtype)
    B, ht, wd = disps.shape
    intr_b = _c(np.broadcast_to(np.asarray(intr_b, dtype).reshape(-1, 4), (B, 4)), dtype)
    ii, jj = _c(ii, np.int64), _c(jj, np.int64)
    coords = np.zeros((len(ii), ht, wd, 2), dtype)
    valid = np.zeros((len(ii), ht, wd, 1), dtype)
    getattr(lib(), "oracle_reproject" + _sfx(dtype))(
        _p(poses), _p(disps), _p(intr_b), _p(ii), _p(jj), ctypes.c_int(len(ii)), ctypes.c_int(ht),
        ctypes.c_int(wd), _p(coords), _p(valid))
    return coords, valid


# ---- correlation --------------------------------------------------------------------------

def _is_half(a):
    return np.asarray(a).dtype == np.float16


def corr_volume(fmap1, fmap2):
    """CorrBlock.corr (corr.py:63-71): fmap [n, C, h, w] -> [n, h1, w1, h2, w2]."""
    n, C, h1, w1 = fmap1.shape
    _, _, h2, w2 = fmap2.shape
    if _is_half(fmap1):
        a = np.ascontiguousarray(fmap1).view(np.uint16)
        b = np.ascontiguousarray(fmap2).view(np.uint16)
        out = np.zeros((n, h1 * w1, h2 * w2), np.uint16)
        lib().oracle_corr_volume_f16(_p(a), _p(b), ctypes.c_int(n), ctypes.c_int(C),
                                     ctypes.c_int(h1 * w1), ctypes.c_int(h2 * w2), _p(out))
        return out.view(np.float16).reshape(n, h1, w1, h2, w2)
    a, b = _c(fmap1, np.float32), _c(fmap2, np.float32)
    out = np.zeros((n, h1 * w1, h2 * w2), np.float32)
    lib().oracle_corr_volume_f32(_p(a), _p(b), ctypes.c_int(n), ctypes.c_int(C), ctypes.c_int(h1 * w1),
                                 ctypes.c_int(h2 * w2), _p(out))
    return out.reshape(n, h1, w1, h2, w2)


def avg_pool2(vol):
    """F.avg_pool2d(.,2,stride=2) on the trailing plane of [n,h1,w1,h2,w2]."""
    n, h1, w1, h2, w2 = vol.shape
    planes = n * h1 * w1
    if _is_half(vol):
        out = np.zeros((n, h1, w1, h2 // 2, w2 // 2), np.uint16)
        lib().oracle_avg_pool2_f16(_p(np.ascontiguousarray(vol).view(np.uint16)), ctypes.c_size_t(planes),
                                   ctypes.c_int(h2), ctypes.c_int(w2), _p(out))
        return out.view(np.float16)
    out = np.zeros((n, h1, w1, h2 // 2, w2 // 2), np.float32)
    lib().oracle_avg_pool2_f32(_p(_c(vol, np.float32)), ctypes.c_size_t(planes), ctypes.c_int(h2),
                               ctypes.c_int(w2), _p(out))
    return out


def corr_pyramid(fmap1, fmap2, num_levels=4):
    """CorrBlock.__init__ (corr.py:24-38)."""
    pyr = [corr_volume(fmap1, fmap2)]
    for _ in range(num_levels - 1):
        pyr.append(avg_pool2(pyr[-1]))
    return pyr


def corr_index_forward(volume, coords, radius):
    """corr_index_forward_kernel (correlation_kernels.cu:19-70)."""
    n, h1, w1, h2, w2 = volume.shape
    coords = _c(coords, np.float32)
    rd = 2 * radius + 1
    args = (ctypes.c_int(n), ctypes.c_int(h1), ctypes.c_int(w1), ctypes.c_int(h2), ctypes.c_int(w2),
            ctypes.c_int(radius))
    if _is_half(volume):
        out = np.zeros((n, rd, rd, h1, w1), np.uint16)
        lib().oracle_corr_index_forward_f16(_p(np.ascontiguousarray(volume).view(np.uint16)), _p(coords),
                                            _p(out), *args)
        return out.view(np.float16)
    out = np.zeros((n, rd, rd, h1, w1), np.float32)
    lib().oracle_corr_index_forward_f32(_p(_c(volume, np.float32)), _p(coords), _p(out), *args)
    return out


def corr_index_backward(volume_shape, coords, corr_grad, radius):
    n, h1, w1, h2, w2 = volume_shape
    coords = _c(coords, np.float32)
    corr_grad = _c(corr_grad, np.float32)
    out = np.zeros(volume_shape, np.float32)
    lib().oracle_corr_index_backward_f32(_p(coords), _p(corr_grad), _p(out), ctypes.c_int(n),
                                         ctypes.c_int(h1), ctypes.c_int(w1), ctypes.c_int(h2),
                                         ctypes.c_int(w2), ctypes.c_int(radius))
    return out


def corr_lookup_pyramid(pyr, coords, radius):
    """CorrBlock.__call__ (corr.py:40-50): coords [n, h, w, 2] -> [n, L*rd*rd, h, w]."""
    coords = np.asarray(coords, np.float32)
    n, h, w, _ = coords.shape
    c = np.ascontiguousarray(coords.transpose(0, 3, 1, 2))
    outs = []
    for lvl, vol in enumerate(pyr):
        o = corr_index_forward(vol, (c / np.float32(2 ** lvl)).astype(np.float32), radius)
        outs.append(o.reshape(n, -1, h, w))
    return np.concatenate(outs, axis=1)


def altcorr_forward(fmap1, fmap2, coords, radius):
    """altcorr_forward_kernel (altcorr_kernel.cu:27-149): float32, or c10::Half arithmetic when the maps are float16."""
    if _is_half(fmap1):
        f1 = np.ascontiguousarray(fmap1).view(np.uint16)
        f2 = np.ascontiguousarray(fmap2, np.float16).view(np.uint16)
        coords = _c(coords, np.float32)
        B, H1, W1, C = f1.shape
        _, H2, W2, _ = f2.shape
        S = coords.shape[1]
        rd = 2 * radius + 1
        out = np.zeros((B, S, rd * rd, H1, W1), np.uint16)
        lib().oracle_altcorr_forward_f16(_p(f1), _p(f2), _p(coords), _p(out), ctypes.c_int(B), ctypes.c_int(S),
                                         ctypes.c_int(H1), ctypes.c_int(W1), ctypes.c_int(H2), ctypes.c_int(W2),
                                         ctypes.c_int(C), ctypes.c_int(radius))
        return out.view(np.float16)
    fmap1, fmap2, coords = _c(fmap1, np.float32), _c(fmap2, np.float32), _c(coords, np.float32)
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    rd = 2 * radius + 1
    out = np.zeros((B, S, rd * rd, H1, W1), np.float32)
    lib().oracle_altcorr_forward_f32(_p(fmap1), _p(fmap2), _p(coords), _p(out), ctypes.c_int(B),
                                     ctypes.c_int(S), ctypes.c_int(H1), ctypes.c_int(W1), ctypes.c_int(H2),
                                     ctypes.c_int(W2), ctypes.c_int(C), ctypes.c_int(radius))
    return out


def altcorr_backward(fmap1, fmap2, coords, corr_grad, radius):
    """altcorr_backward_kernel (altcorr_kernel.cu:152-286), float32 -> (fmap1_grad, fmap2_grad)."""
    fmap1, fmap2, coords = _c(fmap1, np.float32), _c(fmap2, np.float32), _c(coords, np.float32)
    corr_grad = _c(corr_grad, np.float32)
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    g1 = np.zeros_like(fmap1)
    g2 = np.zeros_like(fmap2)
    lib().oracle_altcorr_backward_f32(_p(fmap1), _p(fmap2), _p(coords), _p(corr_grad), _p(g1), _p(g2),
                                      ctypes.c_int(B), ctypes.c_int(S), ctypes.c_int(H1), ctypes.c_int(W1),
                                      ctypes.c_int(H2), ctypes.c_int(W2), ctypes.c_int(C), ctypes.c_int(radius))
    return g1, g2


# ---- pinning hooks (used only by tests/test_oracle_golden.py) -------------------------------------------

def schur_rows(E, C, w, ii, jj, B, ht, wd, t0, t1, A, b, dx=None, dtype=np.float64):
    """step_schur (+ step_backsub) of the restatement on caller-supplied blocks; returns (A - EQE^T, b - EQw, dz)."""
    E, C, w = _c(E, dtype), _c(C, dtype), _c(w, dtype)
    ii, jj = _c(ii, np.int64), _c(jj, np.int64)
    A = _c(A, np.float64).copy()
    b = _c(b, np.float64).copy()
    M, HW = C.shape[0], ht * wd
    dz = np.zeros((M, HW), dtype)
    dxa = _c(dx, dtype) if dx is not None else None
    fn = getattr(lib(), "oracle_schur_rows" + _sfx(dtype))
    fn.restype = ctypes.c_int
    m = fn(_p(E), _p(C), _p(w), _p(ii), _p(jj), ctypes.c_int(len(ii)), ctypes.c_int(B), ctypes.c_int(ht),
           ctypes.c_int(wd), ctypes.c_int(t0), ctypes.c_int(t1), _p(A), _p(b), _p(dxa), _p(dz))
    assert m == M, "depth-block row count mismatch: oracle %d, caller %d" % (m, M)
    return A, b, dz


def sys_solve(A, b, lm, ep, dtype=np.float64):
    A, b = _c(A, np.float64), _c(b, np.float64)
    x = np.zeros_like(b)
    fn = getattr(lib(), "oracle_sys_solve" + _sfx(dtype))
    fn.restype = ctypes.c_int
    ok = fn(_p(A), _p(b), ctypes.c_int(b.shape[0]), ctypes.c_double(lm), ctypes.c_double(ep), _p(x))
    return x, bool(ok)


def _presystem(self, alpha):
    n = 6 * self.P
    A = np.zeros((n, n), np.float64)
    v = np.zeros((n,), np.float64)
    C = np.zeros((self.M, self.ht * self.wd), self.dtype)
    getattr(self.L, "oracle_bacore_presystem" + _sfx(self.dtype))(self.h, _real(self.dtype)(alpha), _p(A), _p(v), _p(C))
    return A, v, C


BACore.presystem = _presystem
