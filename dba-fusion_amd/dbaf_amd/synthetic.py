"""Seeded synthetic keyframe windows for tests and bench.py.

No dataset or network weights exist on the build or GPU boxes, so throughput and
parity are measured on synthetic windows with the shape of the reference's real
workloads: a sliding window of keyframes (poses world->camera as (t, q_xyzw), inverse
depth maps at 1/8 resolution), a covisibility edge set (ii -> jj), per-edge
reprojection targets + confidence weights as produced by the update operator
(/root/reference/dbaf/covisible_graph.py:214-342), and fp16 feature maps for the
correlation volume (/root/reference/dbaf/depth_video.py:64).

Pure numpy (float64 internally) -- this module is product-side input generation and
must not depend on oracle/.
"""
import numpy as np

TUMVI_INTRINSICS_8 = (23.872, 23.872, 31.866, 32.112)  # calib/tumvi.txt (fx fy cx cy) / 8


# ---- minimal SE3 helpers (t, q_xyzw), float64 ------------------------------------------

def _qmul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], -1)


def _qrot(q, v):
    qv = q[..., :3]
    uv = 2.0 * np.cross(qv, v)
    return v + q[..., 3:4] * uv + np.cross(qv, uv)


def se3_exp(xi):
    """xi = (tau, phi) -> (t, q).  T = Exp(xi)."""
    xi = np.asarray(xi, np.float64)
    tau, phi = xi[..., :3], xi[..., 3:]
    th2 = (phi * phi).sum(-1, keepdims=True)
    th = np.sqrt(th2)
    small = th < 1e-6
    ths = np.where(small, 1.0, th)
    imag = np.where(small, 0.5 - th2 / 48.0, np.sin(0.5 * ths) / ths)
    q = np.concatenate([imag * phi, np.cos(0.5 * th)], -1)
    a = np.where(small, 0.5 - th2 / 24.0, (1 - np.cos(ths)) / np.where(small, 1.0, th2))
    b = np.where(small, 1.0 / 6.0 - th2 / 120.0, (ths - np.sin(ths)) / np.where(small, 1.0, th2 * ths))
    pt = np.cross(phi, tau)
    ppt = np.cross(phi, pt)
    t = tau + a * pt + b * ppt
    return np.concatenate([t, q], -1)


def se3_mul(A, B):
    """(A * B): apply B then A."""
    tA, qA, tB, qB = A[..., :3], A[..., 3:], B[..., :3], B[..., 3:]
    return np.concatenate([_qrot(qA, tB) + tA, _qmul(qA, qB)], -1)


def se3_inv(A):
    q = A[..., 3:] * np.array([-1.0, -1.0, -1.0, 1.0])
    return np.concatenate([-_qrot(q, A[..., :3]), q], -1)


def reproject_np(poses, disps, intr, ii, jj):
    """float64 reprojection ii -> jj of every pixel; returns coords [N,h,w,2], depth Z [N,h,w]."""
    poses = np.asarray(poses, np.float64)
    disps = np.asarray(disps, np.float64)
    fx, fy, cx, cy = [float(v) for v in intr]
    _, h, w = disps.shape
    y, x = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    X = np.stack([(x - cx) / fx, (y - cy) / fy, np.ones_like(x)], -1)  # [h,w,3]
    Gij = se3_mul(poses[jj], se3_inv(poses[ii]))  # [N,7]
    stereo = (np.asarray(ii) == np.asarray(jj))
    Gij[stereo] = np.array([-0.1, 0, 0, 0, 0, 0, 1.0])
    q = Gij[:, None, None, 3:]
    t = Gij[:, None, None, :3]
    d = disps[ii][..., None]
    Xj = _qrot(np.broadcast_to(q, (len(ii), h, w, 4)), np.broadcast_to(X, (len(ii), h, w, 3))) + d * t
    Z = Xj[..., 2]
    Zs = np.where(np.abs(Z) < 1e-6, 1e-6, Z)
    coords = np.stack([fx * Xj[..., 0] / Zs + cx, fy * Xj[..., 1] / Zs + cy], -1)
    return coords, Z


# ---- covisibility graphs -------------------------------------------------------------------

def graph_banded(num_kf, radius, extra=()):
    ii, jj = [], []
    for i in range(num_kf):
        for j in range(num_kf):
            if i != j and abs(i - j) <= radius:
                ii.append(i)
                jj.append(j)
    for (a, b) in extra:
        ii += [a, b]
        jj += [b, a]
    return np.asarray(ii, np.int64), np.asarray(jj, np.int64)


def graph_25_96():
    """25 KF / 96 edges: bidirectional |i-j| in {1,2} (94) + (0,3),(3,0)."""
    return graph_banded(25, 2, extra=[(0, 3)])


def graph_32_122():
    """KITTI-360 shape: 32 KF, |i-j| <= 2 -> 122 edges."""
    return graph_banded(32, 2)


def graph_64_512():
    """64 KF / 512 edges: |i-j| <= 4 (492) + (i,i+5) both ways for i < 10 (20)."""
    return graph_banded(64, 4, extra=[(i, i + 5) for i in range(10)])


def graph_64_weak(world):
    """Weak-scaling family of the 64-KF window (bench.py --scaling weak): every frame has exactly `world` out-edges, to its
    nearest neighbours (+1, -1, +2, -2, ...; frames near the ends reach further on the side that exists), i.e. 64 * world
    edges -- 64 per rank once the source frames are dealt out (8 ranks: the |i-j| <= 4 band of graph_64_512 away from the
    ends, 512 edges)."""
    num_kf = 64
    ii, jj = [], []
    for i in range(num_kf):
        picked, d = [], 1
        while len(picked) < world and d < num_kf:
            for j in (i + d, i - d):
                if 0 <= j < num_kf and len(picked) < world:
                    picked.append(j)
            d += 1
        ii += [i] * len(picked)
        jj += picked
    return np.asarray(ii, np.int64), np.asarray(jj, np.int64)


def _box3(a):
    p = np.pad(a, ((0, 0), (1, 1), (1, 1)), mode="edge")
    out = np.zeros_like(a)
    for dy in range(3):
        for dx in range(3):
            out += p[:, dy:dy + a.shape[1], dx:dx + a.shape[2]]
    return out / 9.0


class Window:
    """One synthetic sliding window, all arrays float32 / int64 numpy, reference layouts."""
    pass


def make_window(ii, jj, num_kf, h=64, w=64, seed=0, intr=TUMVI_INTRINSICS_8, buffer=None,
                t0=1, pose_noise=0.01, disp_noise=0.05, target_noise=0.5, sensor_frac=0.0,
                with_fmaps=False, channels=128):
    """Build a window per SURVEY 8(d): smooth trajectory, low-passed inverse depths,
    targets = reprojection(ground truth) + N(0, target_noise px), weights U(0,1) with 10 % zeros.
    The state handed to BA is the ground truth perturbed (pose_noise on the tangent, relative
    disp_noise on inverse depth) so Gauss-Newton has work to do."""
    rng = np.random.default_rng(seed)
    B = buffer if buffer is not None else num_kf + 1
    ii = np.asarray(ii, np.int64)
    jj = np.asarray(jj, np.int64)
    N = len(ii)
    xi0 = np.array([0.05, 0.0, 0.01, 0.0, 0.01, 0.0])
    poses_gt = np.zeros((B, 7))
    poses_gt[:, 6] = 1.0
    for k in range(num_kf):
        poses_gt[k] = se3_exp(k * xi0 + 0.02 * rng.standard_normal(6) * (k > 0))
    disps_gt = np.ones((B, h, w))
    disps_gt[:num_kf] = _box3(rng.uniform(0.2, 2.0, size=(num_kf, h, w)))
    coords, _ = reproject_np(poses_gt, disps_gt, intr, ii, jj)
    target = coords + target_noise * rng.standard_normal(coords.shape)
    weight = rng.uniform(0.0, 1.0, size=coords.shape)
    weight[rng.uniform(size=coords.shape) < 0.1] = 0.0

    poses = poses_gt.copy()
    for k in range(t0, num_kf):
        poses[k] = se3_mul(se3_exp(pose_noise * rng.standard_normal(6)), poses_gt[k])
    disps = disps_gt.copy()
    disps[:num_kf] *= (1.0 + disp_noise * rng.standard_normal((num_kf, h, w)))
    disps = np.clip(disps, 0.05, None)
    disps_sens = np.zeros((B, h, w))
    if sensor_frac > 0:
        mask = rng.uniform(size=(num_kf, h, w)) < sensor_frac
        disps_sens[:num_kf][mask] = (disps_gt[:num_kf] * (1 + 0.01 * rng.standard_normal((num_kf, h, w))))[mask]

    t1 = num_kf
    kx = np.unique(np.concatenate([np.arange(t0, t1), ii]))
    W = Window()
    W.h, W.w, W.B, W.N, W.num_kf, W.t0, W.t1 = h, w, B, N, num_kf, t0, t1
    W.M = len(kx)
    W.kx = kx
    W.ii, W.jj = ii, jj
    W.intrinsics = np.asarray(intr, np.float32)
    W.poses_gt = poses_gt.astype(np.float32)
    W.disps_gt = disps_gt.astype(np.float32)
    W.poses = poses.astype(np.float32)
    W.disps = disps.astype(np.float32)
    W.disps_sens = disps_sens.astype(np.float32)
    # [N,2,h,w] contiguous, ch0 = u/x, ch1 = v/y (covisible_graph.py:332-333)
    W.target = np.ascontiguousarray(target.transpose(0, 3, 1, 2)).astype(np.float32)
    W.weight = np.ascontiguousarray(weight.transpose(0, 3, 1, 2)).astype(np.float32)
    W.eta = np.full((W.M, h, w), 0.2 * 1e-6 + 1e-7, np.float32)  # .2*damping + EP (:330)
    W.lm, W.ep, W.itrs = 1e-4, 0.1, 2
    if with_fmaps:
        W.fmaps = make_fmaps(B, channels, h, w, seed + 1000)
    return W


def make_fmaps(B, C, h, w, seed=0):
    """fp16 N(0,1) feature maps, 3x3-smoothed so neighbouring pixels correlate: [B, C, h, w]."""
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((B * C, h, w)).astype(np.float32)
    f = _box3(f) * 3.0
    return f.reshape(B, C, h, w).astype(np.float16)


def lookup_coords(W, oob_frac=0.05, seed=0):
    """Lookup coordinates as in real use: reprojection of the current state (coherent taps),
    with a fraction of pixels pushed out of bounds.  [N, h, w, 2] float32 (x, y)."""
    rng = np.random.default_rng(seed + 7)
    coords, _ = reproject_np(W.poses, W.disps, W.intrinsics, W.ii, W.jj)
    push = rng.uniform(size=coords.shape[:3]) < oob_frac
    coords[push] += rng.choice([-1.0, 1.0], size=(int(push.sum()), 2)) * rng.uniform(40, 200, size=(int(push.sum()), 2))
    return coords.astype(np.float32)


def window_25_96(seed=0, **kw):
    ii, jj = graph_25_96()
    return make_window(ii, jj, 25, 64, 64, seed=seed, **kw)


def window_32_122(seed=0, **kw):
    ii, jj = graph_32_122()
    return make_window(ii, jj, 32, 28, 107, seed=seed, **kw)


def window_64_512(seed=0, **kw):
    ii, jj = graph_64_512()
    return make_window(ii, jj, 64, 64, 64, seed=seed, **kw)


def window_64_weak(world, seed=0, **kw):
    ii, jj = graph_64_weak(world)
    return make_window(ii, jj, 64, 64, 64, seed=seed, **kw)


def window_tiny_a(seed=0, **kw):
    """4 KF / 6 edges / 16x16 fixture graph."""
    ii = np.array([0, 1, 1, 2, 2, 3], np.int64)
    jj = np.array([1, 0, 2, 1, 3, 2], np.int64)
    return make_window(ii, jj, 4, 16, 16, seed=seed, intr=(6.0, 6.0, 7.7, 8.1), **kw)


def window_tiny_b(seed=0, **kw):
    """6 KF / 14 edges / 24x32 incl. one stereo edge (ii==jj) and fixed-pose (ii<t0) edges."""
    ii = np.array([0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 0, 2, 3, 5], np.int64)
    jj = np.array([1, 0, 2, 1, 3, 2, 4, 3, 5, 4, 2, 0, 3, 3], np.int64)
    return make_window(ii, jj, 6, 24, 32, seed=seed, intr=(12.0, 11.5, 15.6, 12.2), sensor_frac=0.2, **kw)
