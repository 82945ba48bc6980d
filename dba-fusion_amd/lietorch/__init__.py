"""Minimal `lietorch`-compatible SE3 shim (boundary module, SURVEY.md section 8(b) "second boundary").

The reference imports `lietorch.SE3` (princeton-vl/lietorch 0.2, an un-vendored submodule that is
absent from /root/reference and from this image) at
  dbaf/geom/projective_ops.py:4,103-120   (group product, inv, action on homogeneous points, adjT)
  dbaf/depth_video.py:3,224,233           (lietorch.SE3(self.poses[None]))
  dbaf/covisible_graph.py:289,318-321     (.inv(), .matrix(), *, .translation())
  dbaf/dbaf_frontend.py / motion_filter.py / geom/ba.py:26 (.retr)
This module provides exactly that surface on plain torch tensors (any device / float dtype), so the
reference's Python call sites run unmodified.  Conventions (they match the CUDA device functions in
/root/reference/src/droid_kernels.cu:61-178, which restate the same maths):
  data[..., 7] = (tx, ty, tz, qx, qy, qz, qw); tangent xi = (tau, phi);
  X.retr(xi) = Exp(xi) * X  (left perturbation);  X.adjT(a) = Ad(X)^T a.

The hot loop does not go through this shim on the MI355X path: `dbaf_amd.projective_ops`
replaces the ~15 small torch kernels of `projective_transform` by one HIP kernel.
"""
import torch

__version__ = "0.2+dba_amd_shim"


def _cross(a, b):
    return torch.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                        a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                        a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], dim=-1)


def _qmul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                        aw * by + ay * bw + az * bx - ax * bz,
                        aw * bz + az * bw + ax * by - ay * bx,
                        aw * bw - ax * bx - ay * by - az * bz], dim=-1)


def _qinv(q):
    return q * torch.as_tensor([-1.0, -1.0, -1.0, 1.0], dtype=q.dtype, device=q.device)


def _qrot(q, v):
    """rotate v [...,3] by unit quaternion q [...,4] (same formula as actSO3, droid_kernels.cu:61-71)."""
    qv = q[..., :3]
    uv = 2.0 * _cross(qv, v)
    return v + q[..., 3:4] * uv + _cross(qv, uv)


def _so3_exp(phi):
    th2 = (phi * phi).sum(-1, keepdim=True)
    th = torch.sqrt(th2)
    small = th2 < 1e-8
    ths = torch.where(small, torch.ones_like(th), th)
    imag = torch.where(small, 0.5 - th2 / 48.0 + th2 * th2 / 3840.0, torch.sin(0.5 * ths) / ths)
    real = torch.where(small, 1.0 - th2 / 8.0 + th2 * th2 / 384.0, torch.cos(0.5 * ths))
    return torch.cat([imag * phi, real], dim=-1)


def _so3_log(q):
    qv, qw = q[..., :3], q[..., 3:4]
    n2 = (qv * qv).sum(-1, keepdim=True)
    n = torch.sqrt(n2)
    small = n < 1e-6
    ns = torch.where(small, torch.ones_like(n), n)
    # 2*atan2(n, w)/n with sign handling for w<0
    ang = 2.0 * torch.atan2(n, qw.abs()) * torch.sign(torch.where(qw == 0, torch.ones_like(qw), qw))
    fac = torch.where(small, 2.0 / qw - 2.0 * n2 / (3.0 * qw * qw * qw), ang / ns)
    return fac * qv


class LieGroup:
    manifold_dim = 0
    embedded_dim = 0

    def __init__(self, data):
        if isinstance(data, LieGroup):
            data = data.data
        self.data = data

    # -- tensor-like plumbing ---------------------------------------------------------------
    @property
    def shape(self):
        return self.data.shape[:-1]

    @property
    def device(self):
        return self.data.device

    @property
    def dtype(self):
        return self.data.dtype

    def __repr__(self):
        return "%s: size=%s, device=%s, dtype=%s" % (type(self).__name__, tuple(self.shape), self.device, self.dtype)

    def __getitem__(self, index):
        return type(self)(self.data[index])

    def __setitem__(self, index, item):
        self.data[index] = item.data if isinstance(item, LieGroup) else item

    def __len__(self):
        return self.data.shape[0]

    def view(self, *dims):
        if len(dims) == 1 and isinstance(dims[0], (tuple, list, torch.Size)):
            dims = tuple(dims[0])
        return type(self)(self.data.view(*dims, self.embedded_dim))

    def to(self, *args, **kwargs):
        return type(self)(self.data.to(*args, **kwargs))

    def cpu(self):
        return type(self)(self.data.cpu())

    def cuda(self):
        return type(self)(self.data.cuda())

    def float(self):
        return type(self)(self.data.float())

    def double(self):
        return type(self)(self.data.double())

    def detach(self):
        return type(self)(self.data.detach())

    def clone(self):
        return type(self)(self.data.clone())

    def vec(self):
        return self.data

    @property
    def tangent_shape(self):
        return self.data.shape[:-1] + (self.manifold_dim,)


class SE3(LieGroup):
    manifold_dim = 6
    embedded_dim = 7
    id_elem = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0)

    # -- constructors -----------------------------------------------------------------------
    @classmethod
    def Identity(cls, *batch_shape, **kwargs):
        if len(batch_shape) == 1 and isinstance(batch_shape[0], (tuple, list, torch.Size)):
            batch_shape = tuple(batch_shape[0])
        data = torch.as_tensor(cls.id_elem, **kwargs)
        return cls(data.repeat(*batch_shape, 1) if batch_shape else data)

    @classmethod
    def IdentityLike(cls, G):
        return cls.Identity(G.shape, device=G.data.device, dtype=G.data.dtype)

    @classmethod
    def InitFromVec(cls, data):
        return cls(data)

    @classmethod
    def exp(cls, xi):
        tau, phi = xi[..., :3], xi[..., 3:]
        q = _so3_exp(phi)
        th2 = (phi * phi).sum(-1, keepdim=True)
        th = torch.sqrt(th2)
        small = th < 1e-4
        ths = torch.where(small, torch.ones_like(th), th)
        a = torch.where(small, 0.5 - th2 / 24.0, (1 - torch.cos(ths)) / (ths * ths))
        b = torch.where(small, 1.0 / 6.0 - th2 / 120.0, (ths - torch.sin(ths)) / (ths * ths * ths))
        pt = _cross(phi, tau)
        ppt = _cross(phi, pt)
        t = tau + a * pt + b * ppt
        return cls(torch.cat([t, q], dim=-1))

    Exp = exp

    # -- group operations -------------------------------------------------------------------
    def inv(self):
        t, q = self.data[..., :3], self.data[..., 3:]
        qi = _qinv(q)
        return SE3(torch.cat([-_qrot(qi, t), qi], dim=-1))

    def mul(self, other):
        tA, qA = self.data[..., :3], self.data[..., 3:]
        tB, qB = other.data[..., :3], other.data[..., 3:]
        qA, qB = torch.broadcast_tensors(qA, qB)
        tA, tB = torch.broadcast_tensors(tA, tB)
        return SE3(torch.cat([_qrot(qA, tB) + tA, _qmul(qA, qB)], dim=-1))

    def act(self, p):
        t, q = self.data[..., :3], self.data[..., 3:]
        shp = torch.broadcast_shapes(q.shape[:-1], p.shape[:-1])
        qb = q.expand(*shp, 4)
        tb = t.expand(*shp, 3)
        pb = p.expand(*shp, p.shape[-1])
        if p.shape[-1] == 3:
            return _qrot(qb, pb) + tb
        # homogeneous [X, Y, Z, W] -> [R X + W t, W]   (actSE3, droid_kernels.cu:73-80)
        xyz = _qrot(qb, pb[..., :3]) + pb[..., 3:4] * tb
        return torch.cat([xyz, pb[..., 3:4]], dim=-1)

    def __mul__(self, other):
        if isinstance(other, LieGroup):
            return self.mul(other)
        if isinstance(other, torch.Tensor):
            return self.act(other)
        return NotImplemented

    def adjT(self, a):
        """Ad(X)^T a for a [...,6] = (a_tau, a_phi)   (adjSE3, droid_kernels.cu:82-97)."""
        t, q = self.data[..., :3], self.data[..., 3:]
        shp = torch.broadcast_shapes(q.shape[:-1], a.shape[:-1])
        qi = _qinv(q).expand(*shp, 4)
        tb = t.expand(*shp, 3)
        ab = a.expand(*shp, 6)
        at, ap = ab[..., :3], ab[..., 3:]
        return torch.cat([_qrot(qi, at), _qrot(qi, ap) + _qrot(qi, _cross(at, tb))], dim=-1)

    def adj(self, a):
        """Ad(X) a."""
        t, q = self.data[..., :3], self.data[..., 3:]
        shp = torch.broadcast_shapes(q.shape[:-1], a.shape[:-1])
        qb = q.expand(*shp, 4)
        tb = t.expand(*shp, 3)
        ab = a.expand(*shp, 6)
        Rp = _qrot(qb, ab[..., 3:])
        return torch.cat([_qrot(qb, ab[..., :3]) + _cross(tb, Rp), Rp], dim=-1)

    def retr(self, xi):
        """left retraction Exp(xi) * X   (retrSE3, droid_kernels.cu:922-940)."""
        return SE3.exp(xi).mul(self)

    def log(self):
        t, q = self.data[..., :3], self.data[..., 3:]
        phi = _so3_log(q)
        th2 = (phi * phi).sum(-1, keepdim=True)
        th = torch.sqrt(th2)
        small = th < 1e-4
        ths = torch.where(small, torch.ones_like(th), th)
        # V^-1 = I - 1/2 [phi]x + c [phi]x^2
        half = 0.5 * ths
        c = torch.where(small, 1.0 / 12.0 + th2 / 720.0,
                        (1.0 - half * torch.cos(half) / torch.sin(half)) / (ths * ths))
        pt = _cross(phi, t)
        ppt = _cross(phi, pt)
        tau = t - 0.5 * pt + c * ppt
        return torch.cat([tau, phi], dim=-1)

    def matrix(self):
        t, q = self.data[..., :3], self.data[..., 3:]
        eye = torch.eye(3, dtype=q.dtype, device=q.device).expand(*q.shape[:-1], 3, 3)
        R = torch.stack([_qrot(q, eye[..., :, k]) for k in range(3)], dim=-1)
        top = torch.cat([R, t.unsqueeze(-1)], dim=-1)
        bot = torch.zeros(*q.shape[:-1], 1, 4, dtype=q.dtype, device=q.device)
        bot[..., 0, 3] = 1.0
        return torch.cat([top, bot], dim=-2)

    def translation(self):
        """X acting on the origin as a homogeneous point: [t, 1] (lietorch semantics)."""
        t = self.data[..., :3]
        return torch.cat([t, torch.ones_like(t[..., :1])], dim=-1)

    def scale(self, s):
        t, q = self.data[..., :3], self.data[..., 3:]
        return SE3(torch.cat([t * s, q], dim=-1))


class Sim3(LieGroup):
    """Placeholder so `from lietorch import SE3, Sim3` resolves (projective_ops.py:4); the
    DBA-Fusion runtime never constructs a Sim3."""
    manifold_dim = 7
    embedded_dim = 8

    def __init__(self, *a, **k):
        raise NotImplementedError("Sim3 is not part of the DBA hot path")


class SO3(LieGroup):
    manifold_dim = 3
    embedded_dim = 4

    def __init__(self, *a, **k):
        raise NotImplementedError("SO3 is not part of the DBA hot path")


def cat(group_list, dim):
    return type(group_list[0])(torch.cat([g.data for g in group_list], dim=dim))


def stack(group_list, dim):
    return type(group_list[0])(torch.stack([g.data for g in group_list], dim=dim))
