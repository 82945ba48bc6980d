"""Generates tests/golden/*.npz by IMPORTING the reference's own Python in the authoring container.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
(needs /root/reference; the GPU box does not have it, which is why the vectors are committed).

What is captured (SURVEY.md section 8(c) "oracle plan" items 2-3) -- data only, no reference source:
  corr_pyramid.npz   CorrBlock(fmap1, fmap2).corr_pyramid          dbaf/modules/corr.py:24-38,63-71
  schur_solve.npz    geom.chol.schur_solve / block_solve           dbaf/geom/chol.py:32-73
  pinhole.npz        pops.coords_grid / iproj / proj               dbaf/geom/projective_ops.py:11-65
  projective.npz     pops.projective_transform(jacobian=True)      dbaf/geom/projective_ops.py:96-125
  ba_step.npz        geom.ba.BA (one full Gauss-Newton step)       dbaf/geom/ba.py:29-104
  ba2gtsam.npz       BA2GTSAM (numpy twin of the GTSAM fork's op)  dbaf/depth_video.py:20-29
schur_solve.npz and ba_step.npz also hold the reduced system (S, v) that reaches the reference's Cholesky
(captured by wrapping geom.chol.CholeskySolver.apply -- the values are the reference's own).
The native CUDA path (src/*.cu) cannot be built or run here (no nvcc / Eigen / NVIDIA device), and
lietorch / torch_scatter / droid_backends are absent: `lietorch` resolves to this repo's SE3 shim,
the other two to empty stubs (none of the captured functions call into them).
projective_ops.py:105 hard-codes device="cuda"; torch.as_tensor is wrapped to drop that argument.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference/dbaf"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "dba-fusion_amd"))  # lietorch shim
sys.path.insert(0, REF)
for name in ("droid_backends", "torch_scatter"):
    sys.modules[name] = types.ModuleType(name)


def _scatter_sum(src, index, dim=-1, dim_size=None):
    """stand-in for the absent third-party torch_scatter.scatter_sum (rusty1s/pytorch_scatter): out[index[i]] += src[i]"""
    shape = list(src.shape)
    shape[dim] = int(dim_size)
    return torch.zeros(shape, dtype=src.dtype).index_add_(dim, index, src)


sys.modules["torch_scatter"].scatter_sum = _scatter_sum
sys.modules["torch_scatter"].scatter_mean = None

_as_tensor = torch.as_tensor


def _as_tensor_cpu(*a, **k):
    k.pop("device", None)
    return _as_tensor(*a, **k)


torch.as_tensor = _as_tensor_cpu

import geom.projective_ops as pops  # noqa: E402
import geom.chol as chol  # noqa: E402
import geom.ba as rba  # noqa: E402
from geom.chol import schur_solve, block_solve  # noqa: E402
from modules.corr import CorrBlock  # noqa: E402
from lietorch import SE3  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "dba-fusion_amd"))
from dbaf_amd import synthetic as syn  # noqa: E402


def gen_corr():
    g = torch.Generator().manual_seed(0)
    out = {}
    for tag, (n, C, h, w) in {"a": (1, 128, 16, 16), "b": (1, 32, 16, 24)}.items():
        f1 = torch.randn(1, n, C, h, w, generator=g)
        f2 = torch.randn(1, n, C, h, w, generator=g)
        cb = CorrBlock(f1, f2, num_levels=4, radius=3)
        out[f"{tag}_fmap1"] = f1[0].numpy()
        out[f"{tag}_fmap2"] = f2[0].numpy()
        for l, p in enumerate(cb.corr_pyramid):
            out[f"{tag}_lvl{l}"] = p.numpy()
    np.savez_compressed(os.path.join(HERE, "corr_pyramid.npz"), **out)


class _CaptureCholesky:
    """records the (S, v) the reference hands to its Cholesky (chol.py:66-68) and forwards the call"""

    def __init__(self):
        self.calls = []
        self._orig = chol.CholeskySolver.apply

    def __enter__(self):
        def apply(H, b):
            self.calls.append((H.detach().clone(), b.detach().clone()))
            return self._orig(H, b)
        chol.CholeskySolver.apply = staticmethod(apply)
        return self

    def __exit__(self, *a):
        chol.CholeskySolver.apply = self._orig


def gen_schur():
    g = torch.Generator().manual_seed(1)
    B, P, M, D, HW = 1, 3, 4, 6, 20
    out = {}
    for tag in ("", "z_"):  # "z_": the first pose has no depth coupling (E[0] = 0), see test_oracle_golden.py
        J = torch.randn(B, P * D, 64, generator=g, dtype=torch.float64)
        H = (J @ J.transpose(1, 2)).view(B, P, D, P, D).permute(0, 1, 3, 2, 4).contiguous()
        E = 0.1 * torch.randn(B, P, M, D, HW, generator=g, dtype=torch.float64)
        if tag:
            E[:, 0] = 0
        C = 1.0 + torch.rand(B, M, HW, generator=g, dtype=torch.float64)
        v = torch.randn(B, P, D, generator=g, dtype=torch.float64)
        w = torch.randn(B, M, HW, generator=g, dtype=torch.float64)
        with _CaptureCholesky() as cap:
            dx, dz = schur_solve(H, E, C, v, w, ep=0.1, lm=1e-4)
        S, vred = cap.calls[0]
        dxb = block_solve(H, v, ep=0.1, lm=1e-4)
        out.update({tag + "H": H.numpy(), tag + "E": E.numpy(), tag + "C": C.numpy(), tag + "v": v.numpy(),
                    tag + "w": w.numpy(), tag + "dx": dx.numpy(), tag + "dz": dz.numpy(),
                    tag + "dx_block": dxb.numpy(), tag + "S": S.numpy(), tag + "vred": vred.numpy()})
    np.savez_compressed(os.path.join(HERE, "schur_solve.npz"), ep=0.1, lm=1e-4, **out)


def gen_ba_step():
    """One full Gauss-Newton step of the reference's torch BA (geom/ba.py:29-104) on the 4-KF fixture window
    (every transformed depth > 0.25, no depth measurements: the conditions under which it and the CUDA path
    assemble the same system).  Captured: the arguments of schur_solve (= the assembled
    H, E, C, v, w), the reduced system that reaches the Cholesky, dx, dz and the retracted state."""
    W = syn.window_tiny_a(3)
    K = W.num_kf  # the torch BA optimises every row of `poses` from fixedp on: hand it the window only
    poses = SE3(torch.from_numpy(W.poses[:K])[None])
    disps = torch.from_numpy(W.disps[:K])[None]
    intr = torch.from_numpy(np.tile(W.intrinsics, (K, 1)))[None]
    ii, jj = torch.from_numpy(W.ii), torch.from_numpy(W.jj)
    target = torch.from_numpy(W.target).permute(0, 2, 3, 1)[None].contiguous()
    weight = torch.from_numpy(W.weight).permute(0, 2, 3, 1)[None].contiguous()
    kx = torch.unique(ii)
    eta = torch.from_numpy(W.eta)[None][:, :len(kx)]
    rec = {}
    orig = rba.schur_solve

    def spy(H, E, C, v, w, **k):
        rec.update(H=H.clone(), E=E.clone(), C=C.clone(), v=v.clone(), w=w.clone())
        dx, dz = orig(H, E, C, v, w, **k)
        rec.update(dx=dx.clone(), dz=dz.clone())
        return dx, dz

    rba.schur_solve = spy
    try:
        with _CaptureCholesky() as cap:
            poses1, disps1 = rba.BA(target, weight, eta, poses, disps, intr, ii, jj, fixedp=W.t0)
    finally:
        rba.schur_solve = orig
    S, vred = cap.calls[0]
    np.savez_compressed(os.path.join(HERE, "ba_step.npz"), seed=3, t0=W.t0, kx=kx.numpy(),
                        H=rec["H"][0].numpy(), E=rec["E"][0].numpy(), C=rec["C"][0].numpy(), v=rec["v"][0].numpy(),
                        w=rec["w"][0].numpy(), S=S[0].numpy(), vred=vred[0].numpy(), dx=rec["dx"][0].numpy(),
                        dz=rec["dz"][0].numpy(), poses1=poses1.data[0].numpy(), disps1=disps1[0].numpy())


def gen_pinhole():
    g = torch.Generator().manual_seed(2)
    ht, wd = 6, 9
    disps = 0.2 + torch.rand(1, 2, ht, wd, generator=g)
    intr = torch.tensor([[[7.5, 7.1, 4.4, 2.9], [7.5, 7.1, 4.4, 2.9]]])
    grid = pops.coords_grid(ht, wd)
    pts, _ = pops.iproj(disps, intr)
    Xs = torch.randn(1, 2, ht, wd, 4, generator=g)
    Xs[..., 2] = Xs[..., 2].abs() * 2.0  # some below 0.1 exercise the Z clamp (:44)
    xy, _ = pops.proj(Xs, intr)
    xyd, _ = pops.proj(Xs, intr, return_depth=True)
    np.savez_compressed(os.path.join(HERE, "pinhole.npz"), disps=disps.numpy(), intr=intr.numpy(),
                        grid=grid.numpy(), pts=pts.numpy(), Xs=Xs.numpy(), xy=xy.numpy(), xyd=xyd.numpy())


def gen_projective():
    out = {}
    for tag, W in {"a": syn.window_tiny_a(3), "b": syn.window_tiny_b(4)}.items():
        # fp32 like the runtime (projective_ops.py:105 writes a float32 literal into Gij.data)
        poses = SE3(torch.from_numpy(W.poses)[None])
        disps = torch.from_numpy(W.disps)[None]
        intr = torch.from_numpy(np.tile(W.intrinsics, (W.B, 1)))[None]
        ii, jj = torch.from_numpy(W.ii), torch.from_numpy(W.jj)
        coords, valid, (Ji, Jj, Jz) = pops.projective_transform(poses, disps, intr, ii, jj, jacobian=True)
        coords32, valid32 = pops.projective_transform(poses, disps, intr, ii, jj)
        out.update({f"{tag}_poses": W.poses, f"{tag}_disps": W.disps, f"{tag}_intr": W.intrinsics,
                    f"{tag}_ii": W.ii, f"{tag}_jj": W.jj,
                    f"{tag}_coords": coords32[0].numpy(), f"{tag}_valid": valid32[0].numpy()})
        if tag == "a":  # jacobians only for the small window (keeps the fixture small)
            out.update({f"{tag}_Ji": Ji[0].numpy(), f"{tag}_Jj": Jj[0].numpy(), f"{tag}_Jz": Jz[0].numpy()})
    np.savez_compressed(os.path.join(HERE, "projective.npz"), **out)


def gen_ba2gtsam():
    """depth_video.py imports gtsam / multi_sensor / droid_net at module level (absent here), so its BA2GTSAM --
    ten lines of numpy that only need `Tbc.inverse().AdjointMap()` -- is executed from the reference file's own text,
    located by its `def` line, with a small Pose3 stand-in for the (absent) gtsam.Pose3."""
    import ast
    import textwrap
    src = open(os.path.join(REF, "depth_video.py")).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "BA2GTSAM")
    fn.args.args[2].annotation = None  # `Tbc: gtsam.Pose3`
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"np": np}
    exec(compile(ast.fix_missing_locations(mod), "depth_video.py:BA2GTSAM", "exec"), ns)

    class Pose3:  # gtsam::Pose3 semantics: AdjointMap = [[R, 0], [[t]x R, R]] (rotation, translation order)
        def __init__(self, M):
            self.M = np.asarray(M, np.float64)

        def inverse(self):
            return Pose3(np.linalg.inv(self.M))

        def AdjointMap(self):
            R, t = self.M[:3, :3], self.M[:3, 3]
            tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
            return np.block([[R, np.zeros((3, 3))], [tx @ R, R]])

    rng = np.random.default_rng(4)
    P = 4
    M = rng.standard_normal((6 * P, 6 * P + 2))
    H, v = M @ M.T, rng.standard_normal(6 * P)
    q = np.array([0.1, -0.2, 0.3, 0.9])
    q /= np.linalg.norm(q)
    x, y, z, w = q
    T = np.eye(4)
    T[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                 [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                 [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]]
    T[:3, 3] = [0.05, -0.02, 0.11]
    Hg, vg = ns["BA2GTSAM"](H, v, Pose3(T))
    np.savez_compressed(os.path.join(HERE, "ba2gtsam.npz"), H=H, v=v, Tbc=T, Hg=Hg, vg=vg, Ad=Pose3(T).AdjointMap())


if __name__ == "__main__":
    gen_corr()
    gen_schur()
    gen_pinhole()
    gen_projective()
    gen_ba_step()
    gen_ba2gtsam()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
