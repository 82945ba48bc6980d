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
import torch.nn.functional as F

from . import _lib


def _ptr(x):
    return ctypes.c_void_p(x.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class CorrBlock:
    """layout="sheared" (default when the shapes allow it) keeps every level flow-aligned,
    Vs_l[n, dy, dx, pixel] with pixel = y1 * w1 + x1 and the pixel axis padded to a multiple of 64
    (csrc/corr_sheared.hip), so the lookup fetches full cache lines for any map size;
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
        self.h1, self.w1 = int(h1), int(w1)
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
        """one-pass MFMA build of the sheared pyramid (any map up to 128 pixels wide); None if the shape is not
        supported (then: build_pyramid + shear_pyramid)"""
        batch, num, dim, h1, w1 = fmap1.shape
        _, _, _, h2, w2 = fmap2.shape
        lib = _lib.load()
        if not lib.dba_corr_volume_build_sheared_supported(dim, h1, w1, h2, w2, num_levels):
            return None
        n = batch * num
        f1 = fmap1.reshape(n, dim, h1, w1).to(torch.float16).contiguous()
        f2 = fmap2.reshape(n, dim, h2, w2).to(torch.float16).contiguous()
        hw1p = lib.dba_corr_sheared_plane_elems(int(h1), int(w1))
        levels = [torch.empty(n, h2 >> l, w2 >> l, hw1p, dtype=torch.float16, device=f1.device)
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
        """reference-layout levels [n,h1,w1,h2l,w2l] -> flow-aligned levels [n,h2l,w2l,HW1p] (pixel axis padded to a
        multiple of 64; the padding is never read for a real pixel)"""
        lib = _lib.load()
        out = []
        for lvl, v in enumerate(ref_levels):
            n, h1, w1, h2l, w2l = v.shape
            vs = torch.empty(n, h2l, w2l, lib.dba_corr_sheared_plane_elems(int(h1), int(w1)), dtype=v.dtype,
                             device=v.device)
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

    def __call__(self, coords, timing=None):
        """timing = (start, stop): two torch.cuda.Event(enable_timing=True) that have been recorded once (so that they
        exist); they are attached to the lookup kernel's dispatch (sheared layout only), start.elapsed_time(stop) is
        then the kernel's duration -- a measurement hook for bench.py, without marker packets in the stream"""
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
            assert (ht, wd) == (self.h1, self.w1), "coords / volume map size mismatch"
            if timing is not None:
                lib.dba_corr_lookup_arm_timing(ctypes.c_void_p(timing[0].cuda_event), ctypes.c_void_p(timing[1].cuda_event))
            _lib.check(lib.dba_corr_lookup_pyramid_sheared(ptrs, _ptr(c), _ptr(out), n, ht, wd, self.h2, self.w2,
                                                           self.num_levels, self.radius, _stream()),
                       "dba_corr_lookup_pyramid_sheared")
            return out
        dt = _lib.DBA_F16 if vol0.dtype == torch.float16 else _lib.DBA_F32
        _lib.check(lib.dba_corr_lookup_pyramid(ptrs, _ptr(c), _ptr(out), n, ht, wd, int(vol0.shape[3]),
                                               int(vol0.shape[4]), self.num_levels, self.radius, dt, _stream()),
                   "dba_corr_lookup_pyramid")
        return out

    def sheared_level(self, lvl):
        """level `lvl` of the flow-aligned pyramid as [n, h2l, w2l, h1, w1] (a view without the plane padding)"""
        assert self.layout == "sheared"
        v = self.corr_pyramid[lvl]
        return v[..., :self.h1 * self.w1].unflatten(-1, (self.h1, self.w1))

    def cat(self, other):
        for i in range(self.num_levels):
            self.corr_pyramid[i] = torch.cat([self.corr_pyramid[i], other.corr_pyramid[i]], 0)
        return self

    def __getitem__(self, index):
        for i in range(self.num_levels):
            self.corr_pyramid[i] = self.corr_pyramid[i][index]
        return self


class CorrSampler(torch.autograd.Function):
    """dbaf/modules/corr.py:6-20: windowed lookup on a reference-layout volume level, with its adjoint."""

    @staticmethod
    def forward(ctx, volume, coords, radius):
        import droid_backends
        ctx.save_for_backward(volume, coords)
        ctx.radius = radius
        corr, = droid_backends.corr_index_forward(volume, coords, radius)
        return corr

    @staticmethod
    def backward(ctx, grad_output):
        import droid_backends
        volume, coords = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        grad_volume, = droid_backends.corr_index_backward(volume, coords, grad_output, ctx.radius)
        return grad_volume, None, None


class CorrLayer(torch.autograd.Function):
    """dbaf/modules/corr.py:74-88: on-the-fly windowed correlation (altcorr), with its adjoint."""

    @staticmethod
    def forward(ctx, fmap1, fmap2, coords, r):
        import droid_backends
        ctx.r = r
        ctx.save_for_backward(fmap1, fmap2, coords)
        corr, = droid_backends.altcorr_forward(fmap1, fmap2, coords, ctx.r)
        return corr

    @staticmethod
    def backward(ctx, grad_corr):
        import droid_backends
        fmap1, fmap2, coords = ctx.saved_tensors
        grad_corr = grad_corr.contiguous()
        fmap1_grad, fmap2_grad, coords_grad = droid_backends.altcorr_backward(fmap1, fmap2, coords, grad_corr, ctx.r)
        return fmap1_grad, fmap2_grad, coords_grad, None


class AltCorrBlock:
    """dbaf/modules/corr.py:91-139: the memory-saving variant that never materialises the volume; keeps a channels-last
    pyramid of the (pre-scaled) feature maps and correlates on the fly per lookup.  Same constructor, corr_fn and
    __call__ as the reference class; the per-level work is droid_backends.altcorr_forward (csrc/altcorr.hip)."""

    def __init__(self, fmaps, num_levels=4, radius=3):
        self.num_levels = num_levels
        self.radius = radius
        B, N, C, H, W = fmaps.shape
        fmaps = fmaps.view(B * N, C, H, W) / 4.0
        self.pyramid = []
        for i in range(self.num_levels):
            sz = (B, N, H // 2 ** i, W // 2 ** i, C)
            fmap_lvl = fmaps.permute(0, 2, 3, 1).contiguous()
            self.pyramid.append(fmap_lvl.view(*sz))
            fmaps = F.avg_pool2d(fmaps, 2, stride=2)

    def corr_fn(self, coords, ii, jj):
        B, N, H, W, S, _ = coords.shape
        coords = coords.permute(0, 1, 4, 2, 3, 5)
        corr_list = []
        for i in range(self.num_levels):
            fmap1_i = self.pyramid[0][:, ii]
            fmap2_i = self.pyramid[i][:, jj]
            coords_i = (coords / 2 ** i).reshape(B * N, S, H, W, 2).contiguous()
            fmap1_i = fmap1_i.reshape((B * N,) + fmap1_i.shape[2:])
            fmap2_i = fmap2_i.reshape((B * N,) + fmap2_i.shape[2:])
            corr = CorrLayer.apply(fmap1_i.float(), fmap2_i.float(), coords_i, self.radius)
            corr = corr.view(B, N, S, -1, H, W).permute(0, 1, 3, 4, 5, 2)
            corr_list.append(corr)
        return torch.cat(corr_list, dim=2)

    def __call__(self, coords, ii, jj):
        squeeze_output = False
        if len(coords.shape) == 5:
            coords = coords.unsqueeze(dim=-2)
            squeeze_output = True
        corr = self.corr_fn(coords, ii, jj)
        if squeeze_output:
            corr = corr.squeeze(dim=-1)
        return corr.contiguous()
