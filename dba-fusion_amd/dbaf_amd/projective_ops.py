"""Fused reprojection for the update loop (mirror of pops.projective_transform without jacobians,
/root/reference/dbaf/geom/projective_ops.py:96-125, called by DepthVideo.reproject,
/root/reference/dbaf/depth_video.py:221-229): one HIP kernel instead of ~15 torch/lietorch kernels."""
import ctypes

import torch

from . import _lib


def _ptr(x):
    return ctypes.c_void_p(x.data_ptr())


def coords_grid(ht, wd, **kwargs):
    y, x = torch.meshgrid(torch.arange(ht).to(**kwargs).float(), torch.arange(wd).to(**kwargs).float(),
                          indexing="ij")
    return torch.stack([x, y], dim=-1)


def projective_transform(poses, depths, intrinsics, ii, jj):
    """poses: SE3 or tensor [1,B,7]; depths [1,B,ht,wd]; intrinsics [1,B,4]; ii,jj [N] int64.
    Returns coords [1,N,ht,wd,2], valid [1,N,ht,wd,1] (float32)."""
    pdata = poses.data if hasattr(poses, "data") and not isinstance(poses, torch.Tensor) else poses
    pdata = pdata.reshape(-1, 7)
    if not pdata.is_cuda:
        raise RuntimeError("projective_transform (MI355X): HIP device tensors required; no CPU path")
    pdata = pdata.float().contiguous()
    d = depths.reshape(-1, depths.shape[-2], depths.shape[-1]).float().contiguous()
    K = intrinsics.reshape(-1, 4).float().contiguous()
    B, ht, wd = d.shape
    if K.shape[0] == 1:
        K = K.expand(B, 4).contiguous()
    ii = ii.to(device=d.device, dtype=torch.int64).contiguous()
    jj = jj.to(device=d.device, dtype=torch.int64).contiguous()
    N = int(ii.shape[0])
    coords = torch.empty(1, N, ht, wd, 2, dtype=torch.float32, device=d.device)
    valid = torch.empty(1, N, ht, wd, 1, dtype=torch.float32, device=d.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.load().dba_reproject(_ptr(pdata), _ptr(d), _ptr(K), _ptr(ii), _ptr(jj), N, int(ht), int(wd),
                                         _ptr(coords), _ptr(valid), stream), "dba_reproject")
    return coords, valid
