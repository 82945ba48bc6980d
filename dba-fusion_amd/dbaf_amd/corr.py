"""MI355X mirror of the reference's correlation operator surface (dbaf/modules/corr.py).

`CorrBlock(fmap1, fmap2)(coords)`, `.cat(other)` and `[index]` behave like the reference class
(/root/reference/dbaf/modules/corr.py:23-71; call sites dbaf/covisible_graph.py:127-132,224 and
dbaf/motion_filter.py:81), but
  * the all-pairs volume and its 4-level pyramid come from one MFMA kernel chain
    (dba_corr_volume_build) instead of torch.matmul + 3x avg_pool2d;
  * the lookup of all levels is ONE launch that reads the [.., h, w, 2] coords as produced by the
    reprojection and writes the concatenated [1, n, L*(2r+1)^2, h, w] tensor directly
    (dba_corr_lookup_pyramid) instead of 4 launches + permute + torch.cat.
The pyramid tensors keep the reference layout [n, h1, w1, h2>>l, w2>>l], so they remain valid inputs
to droid_backends.corr_index_forward.
"""
import ctypes

import torch

from . import _lib


def _ptr(x):
    return ctypes.c_void_p(x.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class CorrBlock:
    """layout="sheared" (default when the shapes allow it) keeps every level flow-aligned,
    Vs_l[n, dy, dx, y1, x1] (csrc/corr_sheared.hip), so the lookup fetches full cache lines;
    layout="reference" keeps the reference's [n, y1, x1, y2, x2] tensors, which are also valid inputs to
    droid_backends.corr_index_forward.  Both give bit-identical lookups."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=3, layout=None):
        self.num_levels = num_levels
        self.radius = radius
        h1, w1 = fmap1.shape[-2:]
        h2, w2 = fmap2.shape[-2:]
        can_shear = (radius == 3 and (h1, w1) == (h2, w2) and (h2 >> (num_levels - 1)) >= 1
                     and (w2 >> (num_levels - 1)) >= 1)
        if layout is None:
            layout = "sheared" if can_shear else "reference"
        if layout == "sheared" and not can_shear:
            raise RuntimeError("CorrBlock: sheared layout needs radius 3 and equal feature-map sizes")
        self.layout = layout
        self.h2, self.w2 = int(h2), int(w2)
        if layout == "sheared":
            fused = CorrBlock.build_sheared_fused(fmap1, fmap2, num_levels)
            self.corr_pyramid = fused if fused is not None else CorrBlock.shear_pyramid(
                CorrBlock.build_pyramid(fmap1, fmap2, num_levels))
        else:
            self.corr_pyramid = CorrBlock.build_pyramid(fmap1, fmap2, num_levels)

    @staticmethod
    def build_pyramid(fmap1, fmap2, num_levels=4):
        """fmap [batch, num, dim, ht, wd] half on the HIP device -> list of [batch*num, h1, w1, h2>>l, w2>>l]."""
        if not fmap1.is_cuda:
            raise RuntimeError("CorrBlock (MI355X): feature maps must be HIP device tensors; no CPU path")
        batch, num, dim, h1, w1 = fmap1.shape
        _, _, _, h2, w2 = fmap2.shape
        n = batch * num
        f1 = fmap1.reshape(n, dim, h1, w1).to(torch.float16).contiguous()
        f2 = fmap2.reshape(n, dim, h2, w2).to(torch.float16).contiguous()
        lib = _lib.load()
        levels = [torch.empty(n, h1, w1, h2 >> l, w2 >> l, dtype=torch.float16, device=f1.device)
                  for l in range(num_levels)]
        sbytes = lib.dba_corr_volume_scratch_bytes(n, dim, h1, w1, h2, w2)
        scratch = torch.empty(max(sbytes, 1), dtype=torch.uint8, device=f1.device)
        ptrs = (ctypes.c_void_p * num_levels)(*[lv.data_ptr() for lv in levels])
        _lib.check(lib.dba_corr_volume_build(_ptr(f1), _ptr(f2), ptrs, n, dim, h1, w1, h2, w2, num_levels,
                                             _ptr(scratch), sbytes, _stream()), "dba_corr_volume_build")
        return levels

    @staticmethod
    def build_sheared_fused(fmap1, fmap2, num_levels=4):
        """one-pass MFMA build of the sheared pyramid (64-wide maps); None if the shape is not supported"""
        batch, num, dim, h1, w1 = fmap1.shape
        _, _, _, h2, w2 = fmap2.shape
        lib = _lib.load()
        if not lib.dba_corr_volume_build_sheared_supported(dim, h1, w1, h2, w2, num_levels):
            return None
        n = batch * num
        f1 = fmap1.reshape(n, dim, h1, w1).to(torch.float16).contiguous()
        f2 = fmap2.reshape(n, dim, h2, w2).to(torch.float16).contiguous()
        levels = [torch.empty(n, h2 >> l, w2 >> l, h1, w1, dtype=torch.float16, device=f1.device)
                  for l in range(num_levels)]
        sbytes = lib.dba_corr_volume_scratch_bytes(n, dim, h1, w1, h2, w2)
        scratch = torch.empty(max(sbytes, 1), dtype=torch.uint8, device=f1.device)
        ptrs = (ctypes.c_void_p * num_levels)(*[lv.data_ptr() for lv in levels])
        _lib.check(lib.dba_corr_volume_build_sheared(_ptr(f1), _ptr(f2), ptrs, n, dim, h1, w1, h2, w2, num_levels,
                                                     _ptr(scratch), sbytes, _stream()),
                   "dba_corr_volume_build_sheared")
        return levels

    @staticmethod
    def shear_pyramid(ref_levels):
        """reference-layout levels [n,h1,w1,h2l,w2l] -> flow-aligned levels [n,h2l,w2l,h1,w1]"""
        lib = _lib.load()
        out = []
        for lvl, v in enumerate(ref_levels):
            n, h1, w1, h2l, w2l = v.shape
            vs = torch.empty(n, h2l, w2l, h1, w1, dtype=v.dtype, device=v.device)
            _lib.check(lib.dba_corr_shear_level(_ptr(v), _ptr(vs), int(n), int(h1), int(w1), int(h2l), int(w2l),
                                                lvl, _stream()), "dba_corr_shear_level")
            out.append(vs)
        return out

    @staticmethod
    def corr(fmap1, fmap2):
        """all-pairs correlation only (corr.py:63-71) -> [batch, num, ht, wd, ht, wd]"""
        batch, num, dim, ht, wd = fmap1.shape
        lvl0 = CorrBlock.build_pyramid(fmap1, fmap2, 1)[0]
        return lvl0.view(batch, num, ht, wd, fmap2.shape[-2], fmap2.shape[-1])

    def __call__(self, coords):
        batch, num, ht, wd, _ = coords.shape
        n = batch * num
        vol0 = self.corr_pyramid[0]
        assert vol0.shape[0] == n, "coords / volume edge count mismatch"
        c = coords.reshape(n, ht, wd, 2)
        if c.dtype != torch.float32 or not c.is_contiguous():
            c = c.float().contiguous()
        rd = 2 * self.radius + 1
        out = torch.empty(batch, num, self.num_levels * rd * rd, ht, wd, dtype=vol0.dtype, device=vol0.device)
        lib = _lib.load()
        vols = [v if v.is_contiguous() else v.contiguous() for v in self.corr_pyramid]
        ptrs = (ctypes.c_void_p * self.num_levels)(*[v.data_ptr() for v in vols])
        if self.layout == "sheared":
            _lib.check(lib.dba_corr_lookup_pyramid_sheared(ptrs, _ptr(c), _ptr(out), n, ht, wd, self.h2, self.w2,
                                                           self.num_levels, self.radius, _stream()),
                       "dba_corr_lookup_pyramid_sheared")
            return out
        dt = _lib.DBA_F16 if vol0.dtype == torch.float16 else _lib.DBA_F32
        _lib.check(lib.dba_corr_lookup_pyramid(ptrs, _ptr(c), _ptr(out), n, ht, wd, int(vol0.shape[3]),
                                               int(vol0.shape[4]), self.num_levels, self.radius, dt, _stream()),
                   "dba_corr_lookup_pyramid")
        return out

    def cat(self, other):
        for i in range(self.num_levels):
            self.corr_pyramid[i] = torch.cat([self.corr_pyramid[i], other.corr_pyramid[i]], 0)
        return self

    def __getitem__(self, index):
        for i in range(self.num_levels):
            self.corr_pyramid[i] = self.corr_pyramid[i][index]
        return self
