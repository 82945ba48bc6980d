"""MI355X form of the reference's correlation operator surface (dbaf/modules/corr.py).

`CorrBlock(fmap1, fmap2)(coords)`, `.cat(other)` and `[index]` answer like the reference class
(/root/reference/dbaf/modules/corr.py:23-71; call sites dbaf/covisible_graph.py:127-132,166,224 and
dbaf/motion_filter.py:81), but the pyramid is **slot-addressed**:

  * every level lives in ONE store of `capacity` edge volumes (flow-aligned layout, csrc/corr_sheared.hip); which slot
    edge e occupies is a small device table (`slots`) that the lookup kernels read;
  * `CorrBlock(fmap1, fmap2)` does not build anything yet: the volumes are built on first use -- and when the block is
    first used as the argument of `cat`, they are built STRAIGHT INTO FREE SLOTS of the receiving block
    (dba_corr_volume_build_sheared_slots), so `self.corr = self.corr.cat(CorrBlock(...))` (covisible_graph.py:131) moves
    no volume at all where the reference's torch.cat re-writes the whole pyramid (4.3 GB at 96 edges);
  * `corr[mask]` (rm_factors, covisible_graph.py:166) edits the table; the slots of the dropped edges are free again;
  * the all-pairs volume and its 4-level pyramid come from one MFMA kernel (csrc/corr_build_fused.hip) instead of
    torch.matmul + 3x avg_pool2d, the lookup of all levels is ONE launch that writes the concatenated
    [1, n, L*(2r+1)^2, h, w] tensor (instead of 4 launches + permute + torch.cat), and `lookup_reprojected` also takes the
    reprojection (DepthVideo.reproject, depth_video.py:221-229) into that launch.
Lookups are bit-identical to droid_backends.corr_index_forward on the reference-layout pyramid.
"""
import ctypes
import os

import torch
import torch.nn.functional as F

from . import _lib


def _ptr(x):
    return ctypes.c_void_p(x.data_ptr()) if x is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_IDENTITY_SLOTS = {}
_BUILD_SCRATCH = {}
_ONCE_PYRAMID = {}
# CorrBlock(f1, f2)(coords) on a block of at most this many edges that was never built, cat'ed or indexed: the pyramid is built
# into a standing per-stream buffer and looked up in ONE library call, the block itself stays unbuilt (MotionFilter.track builds
# a one-edge block per incoming frame, looks it up once and drops it: motion_filter.py:74-76).  0 switches the path off.
ONCE_MAX_EDGES = int(os.environ.get("DBA_CORR_ONCE_MAX_EDGES", "2"))


def _build_scratch(nbytes, device):
    """the volume build's scratch (the k-block-major copies of maps it cannot read as they lie), kept per (device, stream) and
    only ever grown: builds on one stream are ordered, so they can share it -- an allocation less per build"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _BUILD_SCRATCH.get(key)
    if t is None or t.numel() < nbytes:
        if len(_BUILD_SCRATCH) > 64:
            _BUILD_SCRATCH.clear()
        t = _BUILD_SCRATCH[key] = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
    return t


def _identity_slots(n, device):
    """arange(n) as the slot table of a freshly built block, one tensor per (device, n): slot tables are only ever REPLACED
    (cat / index / compaction make new tensors), never written in place, so blocks can share it -- a launch less per
    CorrBlock(...), which the motion filter constructs once per frame (motion_filter.py:74-76)"""
    key = (device.type, device.index, n)
    t = _IDENTITY_SLOTS.get(key)
    if t is None:
        if len(_IDENTITY_SLOTS) > 256:
            _IDENTITY_SLOTS.clear()
        t = _IDENTITY_SLOTS[key] = torch.arange(n, dtype=torch.int32, device=device)
    return t


class CorrBlock:
    """layout="sheared" (default when the shapes allow it) keeps every level flow-aligned,
    Vs_l[slot, dy, dx, pixel] with pixel = y1 * w1 + x1 and the pixel axis padded to a multiple of 64
    (csrc/corr_sheared.hip), so the lookup fetches full cache lines for any map size;
    layout="reference" keeps the reference's [slot, y1, x1, y2, x2] tensors, which (gathered: `corr_pyramid`) are also
    valid inputs to droid_backends.corr_index_forward.  Both give bit-identical lookups.

    capacity: slots to allocate when the stores are created (default: `CorrBlock.default_capacity`, i.e. the environment
    variable DBA_CORR_SLOTS, or the number of edges of the first build); a `cat` that finds no free slot grows the
    stores by half (one copy of the live pyramid, amortised) -- an integration that knows `max_factors` sets
    `CorrBlock.default_capacity = max_factors` once and never pays it."""

    default_capacity = int(os.environ.get("DBA_CORR_SLOTS", "0"))

    def __init__(self, fmap1, fmap2, num_levels=4, radius=3, layout=None, capacity=None):
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
        if not fmap1.is_cuda:
            raise RuntimeError("CorrBlock (MI355X): feature maps must be HIP device tensors; no CPU path")
        self.layout = layout
        self.h1, self.w1 = int(h1), int(w1)
        self.h2, self.w2 = int(h2), int(w2)
        self.n = int(fmap1.shape[0] * fmap1.shape[1])
        self._capacity_hint = int(capacity if capacity is not None else CorrBlock.default_capacity)
        # not built yet: the maps are kept (with their in-place versions: a build after an in-place write would not be the
        # reference's result any more, so that raises) until first use or until a `cat` absorbs this block
        self._pending = (fmap1, fmap2, fmap1._version, fmap2._version)
        self._stores = None        # per level: [capacity, ...] device tensor
        self._slots = None         # device int32 [n]: slot of edge e
        self._slots_host = None    # list mirror of _slots, or None when stale (after a device-side index)
        self._identity = True      # slots == arange(n) and capacity == n: the stores ARE the pyramid
        self.stats = dict(built_edges=0, copied_edges=0, grown=0)
        self._once_used = False    # the one-call build + lookup has served this (still unbuilt) block

    # ---- construction ------------------------------------------------------------------------------------------------
    @classmethod
    def from_pyramid(cls, levels, layout, radius=3, hw=None):
        """wrap existing level tensors without a build (recorded volumes in tests, tools/replay_dump.py): reference layout
        [n,h1,w1,h2l,w2l], or flow-aligned [n,h2l,w2l,HW1p] with hw = (h1, w1)"""
        self = cls.__new__(cls)
        self.num_levels, self.radius, self.layout = len(levels), radius, layout
        v0 = levels[0]
        if layout == "reference":
            self.n, self.h1, self.w1, self.h2, self.w2 = (int(x) for x in v0.shape)
        else:
            self.n, self.h2, self.w2 = (int(x) for x in v0.shape[:3])
            self.h1, self.w1 = (int(x) for x in (hw if hw is not None else (self.h2, self.w2)))
        self._capacity_hint, self._pending = 0, None
        self._stores = [v if v.is_contiguous() else v.contiguous() for v in levels]
        self._slots = _identity_slots(self.n, v0.device)
        self._slots_host, self._identity = list(range(self.n)), True
        self.stats = dict(built_edges=0, copied_edges=0, grown=0)
        return self

    @classmethod
    def from_reference(cls, ref_levels, radius=3):
        """reference-layout levels -> a flow-aligned block (one re-layout pass per level)"""
        self = cls.from_pyramid(ref_levels, "reference", radius)
        self._stores = CorrBlock.shear_pyramid(self._stores)
        self.layout = "sheared"
        return self

    def _level_shape(self, lvl, cap):
        if self.layout == "sheared":
            hw1p = _lib.load().dba_corr_sheared_plane_elems(self.h1, self.w1)
            return (cap, self.h2 >> lvl, self.w2 >> lvl, hw1p)
        return (cap, self.h1, self.w1, self.h2 >> lvl, self.w2 >> lvl)

    def _materialise(self):
        """build this block's own stores from its pending maps (first use without a `cat` into another block)"""
        if self._pending is None:
            return
        f1, f2 = self._take_pending()
        cap = max(self.n, self._capacity_hint)
        dev = f1.device
        if self.layout == "sheared":
            hw1p = _lib.load().dba_corr_sheared_plane_elems(self.h1, self.w1)
            shapes = [(cap, self.h2 >> l, self.w2 >> l, hw1p) for l in range(self.num_levels)]
        else:
            shapes = [self._level_shape(l, cap) for l in range(self.num_levels)]
        self._stores = [torch.empty(sh, dtype=torch.float16, device=dev) for sh in shapes]
        self._slots_host = list(range(self.n))
        self._slots = _identity_slots(self.n, dev)
        self._identity = (cap == self.n)
        self._build_into(f1, f2, None)

    def build(self):
        """force the build now (a block is otherwise built at its first lookup, or inside the `cat` that absorbs it)"""
        self._materialise()
        return self

    def _take_pending(self):
        f1, f2, v1, v2 = self._pending
        if f1._version != v1 or f2._version != v2:
            raise RuntimeError("CorrBlock: the feature maps were written in place between CorrBlock(...) and the build")
        self._pending = None
        return f1, f2

    def _build_into(self, fmap1, fmap2, out_slots):
        """volumes of the edges (fmap1[k], fmap2[k]) -> slots out_slots[k] (device int32, None: slot k) of self._stores"""
        lib = _lib.load()
        batch, num, dim, h1, w1 = fmap1.shape
        n = batch * num
        if n == 0:
            return
        f1 = fmap1.reshape(n, dim, h1, w1)
        f2 = fmap2.reshape(n, dim, self.h2, self.w2)
        if f1.dtype != torch.float16 or not f1.is_contiguous():   # (the usual case -- half maps as the encoder left them -- costs no
            f1 = f1.to(torch.float16).contiguous()                #  torch call beyond the view: this path runs once per frame)
        if f2.dtype != torch.float16 or not f2.is_contiguous():
            f2 = f2.to(torch.float16).contiguous()
        sbytes = lib.dba_corr_volume_scratch_bytes(n, dim, h1, w1, self.h2, self.w2)
        scratch = _build_scratch(sbytes, f1.device)
        self.stats["built_edges"] += n
        if self.layout == "sheared" and lib.dba_corr_volume_build_sheared_supported(dim, h1, w1, self.h2, self.w2,
                                                                                    self.num_levels):
            ptrs = (ctypes.c_void_p * self.num_levels)(*[s.data_ptr() for s in self._stores])
            _lib.check(lib.dba_corr_volume_build_sheared_slots(_ptr(f1), _ptr(f2), ptrs, _ptr(out_slots), n, dim, h1, w1,
                                                               self.h2, self.w2, self.num_levels, _ptr(scratch), sbytes,
                                                               _stream()), "dba_corr_volume_build_sheared_slots")
            return
        # shapes the fused kernel does not take, and the reference layout: build compactly, then place
        levels = CorrBlock.build_pyramid(fmap1, fmap2, self.num_levels)
        if self.layout == "sheared":
            levels = CorrBlock.shear_pyramid(levels)
        for store, lv in zip(self._stores, levels):
            if out_slots is None:
                store[:n].copy_(lv)
            else:
                store.index_copy_(0, out_slots.long(), lv)

    # ---- the reference's static helpers (compact tensors) --------------------------------------------------------------
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

    # ---- the pyramid as the reference exposes it -----------------------------------------------------------------------
    @property
    def corr_pyramid(self):
        """list of per-level tensors with the edges in order, [n, ...] (layout "reference": the reference's
        [n,h1,w1,h2l,w2l], valid for droid_backends.corr_index_forward).  The stores themselves when the slot table is the
        identity over a full store (a block that was never cat'ed into / indexed), a gathered copy otherwise."""
        self._materialise()
        if self._identity:
            return self._stores
        idx = self._slots.long()
        return [s.index_select(0, idx) for s in self._stores]

    @staticmethod
    def map_pixels(level, h1, w1):
        """a flow-aligned level [..., HW1p] as [..., h1, w1]: the real pixels in row-major order -- without the plane padding,
        un-tiled where the planes keep their pixels in 4 x 16 tiles (dba_corr_sheared_tiled; the last tile of a row may reach
        past the map: those padding pixels are dropped)"""
        lib = _lib.load()
        hg, wg = ctypes.c_int(0), ctypes.c_int(0)
        tw = lib.dba_corr_sheared_grid(int(h1), int(w1), ctypes.byref(hg), ctypes.byref(wg))   # tile width (0: row-major) and the
        if tw:                                                                                 # grid the tiles are counted on
            hg, wg = hg.value, wg.value
            th, tx = 64 // tw, wg // tw
            v = level.unflatten(-1, (hg // th, tx, th, tw)).movedim(-2, -3)      # [..., hg / th, th, tiles_x, tw]
            return v.reshape(v.shape[:-4] + (hg, wg))[..., :h1, :w1]
        return level[..., :h1 * w1].unflatten(-1, (h1, w1))

    def sheared_level(self, lvl):
        """level `lvl` of the flow-aligned pyramid as [n, h2l, w2l, h1, w1] (map_pixels)"""
        assert self.layout == "sheared"
        return CorrBlock.map_pixels(self.corr_pyramid[lvl], self.h1, self.w1)

    @property
    def capacity(self):
        self._materialise()
        return int(self._stores[0].shape[0])

    # ---- lookups -------------------------------------------------------------------------------------------------------
    def _store_ptrs(self):
        return (ctypes.c_void_p * self.num_levels)(*[s.data_ptr() for s in self._stores])

    def _lookup_once(self, coords):
        """the first lookup of a small block that has not been built: build into the stream's standing pyramid buffer + lookup, one
        library call; the block stays unbuilt (a second lookup, a cat or an index builds it the regular way).  None: not this case"""
        f1, f2, v1, v2 = self._pending
        if f1._version != v1 or f2._version != v2:
            return None   # (the regular path raises)
        if f1.dtype != torch.float16 or f2.dtype != torch.float16 or not f1.is_contiguous() or not f2.is_contiguous():
            return None
        batch, num, ht, wd, _ = coords.shape
        n, dim = self.n, int(f1.shape[2])
        if batch * num != n or (ht, wd) != (self.h1, self.w1) or self.radius != 3:
            return None
        lib = _lib.load()
        if not lib.dba_corr_volume_build_sheared_supported(dim, self.h1, self.w1, self.h2, self.w2, self.num_levels):
            return None
        dev = f1.device
        stream = _stream()
        key = (dev.index, stream.value, n, dim, self.h1, self.w1, self.h2, self.w2, self.num_levels)
        ent = _ONCE_PYRAMID.get(key)
        if ent is None:
            if len(_ONCE_PYRAMID) > 8:
                _ONCE_PYRAMID.clear()
            pbytes = lib.dba_corr_once_pyramid_bytes(n, self.h1, self.w1, self.h2, self.w2, self.num_levels)
            sbytes = lib.dba_corr_volume_scratch_bytes(n, dim, self.h1, self.w1, self.h2, self.w2)
            ent = _ONCE_PYRAMID[key] = (torch.empty(pbytes, dtype=torch.uint8, device=dev), pbytes,
                                        torch.empty(max(sbytes, 1), dtype=torch.uint8, device=dev), sbytes)
        c = coords.reshape(n, ht, wd, 2)
        if c.dtype != torch.float32 or not c.is_contiguous():
            c = c.float().contiguous()
        rd = 2 * self.radius + 1
        out = torch.empty(batch, num, self.num_levels * rd * rd, ht, wd, dtype=torch.float16, device=dev)
        _lib.check(lib.dba_corr_build_lookup_once_sheared(f1.data_ptr(), f2.data_ptr(), c.data_ptr(), out.data_ptr(),
                                                          ent[0].data_ptr(), ent[1], ent[2].data_ptr(), ent[3], n, dim, self.h1,
                                                          self.w1, self.h2, self.w2, self.num_levels, self.radius, stream),
                   "dba_corr_build_lookup_once_sheared")
        self._once_used = True
        self.stats["built_edges"] += n
        return out

    def __call__(self, coords, timing=None):
        """timing = (start, stop): two torch.cuda.Event(enable_timing=True) that have been recorded once (so that they
        exist); they are attached to the lookup kernel's dispatch (sheared layout only), start.elapsed_time(stop) is
        then the kernel's duration -- a measurement hook for bench.py, without marker packets in the stream"""
        if self._pending is not None and self.n <= ONCE_MAX_EDGES and not self._once_used and timing is None \
                and self.layout == "sheared":
            out = self._lookup_once(coords)
            if out is not None:
                return out
        self._materialise()
        batch, num, ht, wd, _ = coords.shape
        n = batch * num
        assert n == self.n, "coords / volume edge count mismatch"
        c = coords.reshape(n, ht, wd, 2)
        if c.dtype != torch.float32 or not c.is_contiguous():
            c = c.float().contiguous()
        rd = 2 * self.radius + 1
        v0 = self._stores[0]
        out = torch.empty(batch, num, self.num_levels * rd * rd, ht, wd, dtype=v0.dtype, device=v0.device)
        lib = _lib.load()
        slots = None if self._identity else self._slots
        if self.layout == "sheared":
            assert (ht, wd) == (self.h1, self.w1), "coords / volume map size mismatch"
            if timing is not None:
                lib.dba_corr_lookup_arm_timing(ctypes.c_void_p(timing[0].cuda_event), ctypes.c_void_p(timing[1].cuda_event))
            _lib.check(lib.dba_corr_lookup_pyramid_sheared_slots(self._store_ptrs(), _ptr(slots), _ptr(c), _ptr(out), n, ht,
                                                                 wd, self.h2, self.w2, self.num_levels, self.radius,
                                                                 _stream()), "dba_corr_lookup_pyramid_sheared_slots")
            return out
        dt = _lib.DBA_F16 if v0.dtype == torch.float16 else _lib.DBA_F32
        _lib.check(lib.dba_corr_lookup_pyramid_slots(self._store_ptrs(), _ptr(slots), _ptr(c), _ptr(out), n, ht, wd,
                                                     self.h2, self.w2, self.num_levels, self.radius, dt, _stream()),
                   "dba_corr_lookup_pyramid_slots")
        return out

    def lookup_reprojected(self, poses, disps, intrinsics, ii, jj, timing=None):
        """DepthVideo.reproject(ii, jj) (depth_video.py:221-229 -> pops.projective_transform, projective_ops.py:96-125) and
        CorrBlock.__call__ in ONE launch: the lookup kernel computes its pixels' coordinates in its prologue.
        poses [B,7] (or [1,B,7] / an SE3), disps [B,h,w] (or [1,B,h,w]), intrinsics [B,4] / [1,B,4] / [4]; ii, jj [n].
        Returns (corr [1,n,L*49,h,w], coords [1,n,h,w,2], valid [1,n,h,w,1]) -- corr and coords bit-identical to
        projective_transform followed by __call__."""
        if self.layout != "sheared":
            raise RuntimeError("lookup_reprojected needs the flow-aligned layout")
        self._materialise()
        pdata = poses.data if hasattr(poses, "data") and not isinstance(poses, torch.Tensor) else poses
        pdata = pdata.reshape(-1, 7).float().contiguous()
        d = disps.reshape(-1, disps.shape[-2], disps.shape[-1]).float().contiguous()
        B, ht, wd = d.shape
        K = intrinsics.reshape(-1, 4).float()
        K = (K.expand(B, 4) if K.shape[0] == 1 else K).contiguous()
        ii = ii.to(device=d.device, dtype=torch.int64).contiguous()
        jj = jj.to(device=d.device, dtype=torch.int64).contiguous()
        n = int(ii.shape[0])
        assert n == self.n and (ht, wd) == (self.h1, self.w1), "edge count / map size mismatch"
        coords = torch.empty(1, n, ht, wd, 2, dtype=torch.float32, device=d.device)
        valid = torch.empty(1, n, ht, wd, 1, dtype=torch.float32, device=d.device)
        rd = 2 * self.radius + 1
        out = torch.empty(1, n, self.num_levels * rd * rd, ht, wd, dtype=torch.float16, device=d.device)
        lib = _lib.load()
        if timing is not None:
            lib.dba_corr_lookup_arm_timing(ctypes.c_void_p(timing[0].cuda_event), ctypes.c_void_p(timing[1].cuda_event))
        slots = None if self._identity else self._slots
        _lib.check(lib.dba_corr_lookup_reproject_sheared(self._store_ptrs(), _ptr(slots), _ptr(pdata), _ptr(d), _ptr(K),
                                                         _ptr(ii), _ptr(jj), _ptr(coords), _ptr(valid), _ptr(out), n, ht, wd,
                                                         self.h2, self.w2, self.num_levels, self.radius, _stream()),
                   "dba_corr_lookup_reproject_sheared")
        return out, coords, valid

    # ---- cat / index: table edits --------------------------------------------------------------------------------------
    def _host_slots(self):
        if self._slots_host is None:   # after a device-side index: one small copy (n ints), at the next graph change
            self._slots_host = [int(v) for v in self._slots.cpu().tolist()]
        return self._slots_host

    def _grow(self, need):
        self._materialise()
        cap = int(self._stores[0].shape[0])
        new_cap = max(cap + (cap + 1) // 2, need, self._capacity_hint)
        new = []
        for lvl, s in enumerate(self._stores):
            t = torch.empty(self._level_shape(lvl, new_cap), dtype=s.dtype, device=s.device)
            t[:cap].copy_(s)        # slot numbers stay valid
            new.append(t)
        self._stores = new
        self._identity = False
        self.stats["grown"] += 1

    def cat(self, other):
        """append the edges of `other` (corr.py:52-55).  A block that has not been used yet is built straight into this
        block's free slots; a built one has its volumes copied there (the new edges' bytes only)."""
        assert (other.num_levels, other.radius, other.h1, other.w1, other.h2, other.w2) == (
            self.num_levels, self.radius, self.h1, self.w1, self.h2, self.w2), "CorrBlock.cat: different shapes"
        self._materialise()
        k = other.n
        if k == 0:
            return self
        used = self._host_slots()
        cap = int(self._stores[0].shape[0])
        if len(used) + k > cap:
            self._grow(len(used) + k)
            cap = int(self._stores[0].shape[0])
        taken = set(used)
        free = [s for s in range(cap) if s not in taken][:k]
        new_slots = torch.tensor(free, dtype=torch.int32, device=self._slots.device)
        if other._pending is not None and other.layout == self.layout:
            f1, f2 = other._take_pending()
            self._build_into(f1, f2, new_slots)
        else:
            other._materialise()
            src = other.corr_pyramid
            if other.layout != self.layout:
                raise RuntimeError("CorrBlock.cat: different layouts")
            for store, lv in zip(self._stores, src):
                store.index_copy_(0, new_slots.long(), lv)
            self.stats["copied_edges"] += k
        self._slots = torch.cat([self._slots, new_slots])
        self._slots_host = used + free
        self.n += k
        self._identity = (self._slots_host == list(range(cap)))
        return self

    def __getitem__(self, index):
        """keep / re-order edges (corr.py:57-60; rm_factors passes a boolean mask): the slot table is indexed, nothing
        else moves; dropped edges' slots are found free at the next `cat`.  The table is edited on the host (one small
        device-to-host copy of the index -- the synchronisation torch's own boolean indexing performs anyway) so that the
        next `cat` knows the free slots without asking the device."""
        self._materialise()
        host = self._host_slots()
        if isinstance(index, torch.Tensor):
            if index.dtype == torch.bool:
                keep = index.reshape(-1).cpu().tolist()
                assert len(keep) == len(host), "CorrBlock[mask]: mask length != number of edges"
                new = [s for s, k in zip(host, keep) if k]
            else:
                new = [host[i] for i in index.reshape(-1).cpu().tolist()]
        elif isinstance(index, slice):
            new = host[index]
        elif isinstance(index, int):
            new = [host[index]]
        else:
            new = [host[i] for i in index]
        self._slots_host = list(new)
        self._slots = torch.tensor(self._slots_host, dtype=torch.int32, device=self._slots.device)
        self.n = len(self._slots_host)
        self._identity = False
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
    """The memory-saving correlation of the reference (dbaf/modules/corr.py:91-139; dead at runtime there): no volume, the
    windowed correlation is formed on the fly from the feature maps at every lookup.

    The feature pyramid is kept ONCE, channels-last -- [B, N, H >> l, W >> l, C] of fmaps / 4, level l + 1 the 2x2 average of
    level l -- with float32 twins made at the first lookup (what the reference's per-lookup `.float()` of its half pyramid
    produces, so a lookup converts nothing).  Without autograd a lookup is ONE launch for all levels
    (dba_altcorr_pyramid_forward: the kernel indexes the maps by ii / jj, divides the coordinates by 2^l itself and writes
    every level's 49 channels into its slice of the output: no gathered copies of the maps, no scaled copies of the
    coordinates); when a gradient is wanted the levels go through `CorrLayer` one by one -- same arithmetic, same bits."""

    def __init__(self, fmaps, num_levels=4, radius=3):
        self.num_levels = num_levels
        self.radius = radius
        B, N, C, H, W = fmaps.shape
        level = fmaps.reshape(B * N, C, H, W) / 4.0
        self.pyramid = []
        for lvl in range(num_levels):
            if lvl > 0:
                level = F.avg_pool2d(level, 2, stride=2)
            self.pyramid.append(level.permute(0, 2, 3, 1).reshape(B, N, H >> lvl, W >> lvl, C).contiguous())
        self._f32 = None
        self.mfma = os.environ.get("DBA_ALTCORR_MFMA", "1") != "0"   # half pyramids: the matrix-core form (csrc/altcorr.hip)

    def _float_pyramid(self):
        if self._f32 is None:
            self._f32 = [p if p.dtype == torch.float32 else p.float() for p in self.pyramid]
        return self._f32

    def corr_fn(self, coords, ii, jj):
        """coords [B, N, H, W, S, 2] -> [B, N, L * 49, H, W, S]"""
        B, N, H, W, S, _ = coords.shape
        rd2 = (2 * self.radius + 1) ** 2
        cs = coords.movedim(4, 2).reshape(B * N, S, H, W, 2)               # the S coordinate sets in front of the pixels
        wants_grad = torch.is_grad_enabled() and (coords.requires_grad or any(p.requires_grad for p in self.pyramid))
        F_ = self.pyramid[0].shape[1]                                      # frames per batch entry
        if not wants_grad and cs.is_cuda and (H >> (self.num_levels - 1)) >= 1 and (W >> (self.num_levels - 1)) >= 1 \
                and B * N * S * self.num_levels <= 65535:
            ii_ = torch.as_tensor(ii, device=cs.device).to(torch.int64).reshape(-1)
            jj_ = torch.as_tensor(jj, device=cs.device).to(torch.int64).reshape(-1)
            # (the reference fails on its reshape when the index lists do not match the coordinates; this path would read
            # out of bounds on the device.  The index VALUES are the caller's contract, as in the reference's gather.)
            if ii_.numel() != N or jj_.numel() != N:
                raise RuntimeError("AltCorrBlock: ii / jj must have one entry per edge (%d), got %d / %d"
                                   % (N, ii_.numel(), jj_.numel()))
            if B > 1:   # frame index inside the flattened [B * F] maps
                off = (torch.arange(B, device=cs.device) * F_)[:, None]
                ii_, jj_ = (off + ii_[None]).reshape(-1), (off + jj_[None]).reshape(-1)
            c = cs if (cs.dtype == torch.float32 and cs.is_contiguous()) else cs.float().contiguous()
            out = torch.empty(B * N, S, self.num_levels * rd2, H, W, dtype=torch.float32, device=cs.device)
            lib = _lib.load()
            half = self.pyramid
            Cn = int(half[0].shape[-1])
            if self.mfma and half[0].dtype == torch.float16 and self.radius == 3 and Cn % 16 == 0 and Cn <= 128 \
                    and H * W * Cn < 2 ** 31 - 1:
                # the reference's case (half maps under autocast, `.float()` at the lookup): the halves go to the matrix cores
                # as they are -- exact products, float sums; no float twins of the pyramid are made at all
                ptrs = (ctypes.c_void_p * self.num_levels)(*[p.data_ptr() for p in half])
                _lib.check(lib.dba_altcorr_pyramid_forward_f16maps(_ptr(half[0]), ptrs, _ptr(ii_.contiguous()),
                                                                   _ptr(jj_.contiguous()), _ptr(c), _ptr(out), B * N, S, H, W, Cn,
                                                                   self.num_levels, self.radius, _stream()),
                           "dba_altcorr_pyramid_forward_f16maps")
                return out.reshape(B, N, S, self.num_levels * rd2, H, W).movedim(2, -1)
            pyr = self._float_pyramid()
            ptrs = (ctypes.c_void_p * self.num_levels)(*[p.data_ptr() for p in pyr])
            _lib.check(lib.dba_altcorr_pyramid_forward(_ptr(pyr[0]), ptrs, _ptr(ii_.contiguous()), _ptr(jj_.contiguous()), _ptr(c),
                                                       _ptr(out), B * N, S, H, W, int(pyr[0].shape[-1]), self.num_levels,
                                                       self.radius, _lib.DBA_F32, _stream()), "dba_altcorr_pyramid_forward")
            return out.reshape(B, N, S, self.num_levels * rd2, H, W).movedim(2, -1)
        pyr = self._float_pyramid()
        src = pyr[0][:, ii].reshape(B * N, H, W, -1)                       # level-0 maps of the source frames
        out = coords.new_empty(B, N, self.num_levels * rd2, H, W, S, dtype=src.dtype)
        for lvl, tgt_all in enumerate(pyr):
            tgt = tgt_all[:, jj]
            tgt = tgt.reshape((B * N,) + tuple(tgt.shape[2:]))
            c = CorrLayer.apply(src, tgt, (cs / 2 ** lvl).contiguous(), self.radius)          # [B*N, S, 49, H, W]
            out[:, :, lvl * rd2:(lvl + 1) * rd2] = c.reshape(B, N, S, rd2, H, W).movedim(2, -1)
        return out

    def __call__(self, coords, ii, jj):
        if coords.dim() == 5:   # [B, N, H, W, 2]: one coordinate set
            return self.corr_fn(coords[..., None, :], ii, jj)[..., 0].contiguous()
        return self.corr_fn(coords, ii, jj).contiguous()
