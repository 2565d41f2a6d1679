"""Input preparation in front of the model, on the GPU.

Mirrors the reference's dataset code: `pcd.estimate_normals(search_param=o3d.geometry.KDTreeSearchParamKNN(knn=33))`
followed by `normal_redirect(points, normals, view_point)` (dataset/tdmatch.py:120-127, dataset/fdmatch.py:83-90,
dataset/common.py:312-320).  Open3D (0.13.0, requirements.txt:64) is not part of the reference tree; its algorithm is
restated in csrc/prep.hip.  No CPU fallback.
"""
import ctypes

import torch

from . import _lib as L
from .pointops import GRID_MIN_POINTS


def _vp(view_point):
    v = [float(x) for x in (view_point if view_point is not None else (0.0, 0.0, 0.0))]
    return (ctypes.c_float * 3)(*v)


def estimate_normals(xyz, offset, knn=33, view_point=(0.0, 0.0, 0.0), use_grid=None):
    """xyz (n,3) fp32 device tensor of b concatenated clouds, offset (b) cumulative int32 -> normals (n,3) fp32.

    view_point=None keeps the unoriented PCA direction (Open3D's sign is arbitrary); otherwise the result equals
    normal_redirect(points, open3d_normals, view_point)."""
    if not xyz.is_cuda:
        raise L.RoitrError("roitr_amd.prep needs ROCm device tensors (no CPU fallback)")
    xyz = xyz.contiguous().float()
    offset = offset.to(torch.int32).contiguous()
    n, b = int(xyz.shape[0]), int(offset.shape[0])
    out = torch.empty((n, 3), dtype=torch.float32, device=xyz.device)
    if n == 0:
        return out
    if use_grid is None:
        use_grid = n > GRID_MIN_POINTS * b
    lib = L.lib()
    lib.roitr_normals_workspace_bytes.restype = ctypes.c_size_t
    ws = torch.empty(lib.roitr_normals_workspace_bytes(b, n, int(knn)), dtype=torch.uint8, device=xyz.device)
    vp = _vp(view_point) if view_point is not None else None
    L.check(lib.roitr_estimate_normals(b, n, L.ptr(xyz), L.ptr(offset), int(knn), 1 if use_grid else 0, vp, L.ptr(out), L.ptr(ws),
                                       L.stream_ptr()), "estimate_normals")
    return out


def normal_redirect(points, normals, view_point):
    """dataset/common.py:312-320: make the normals point towards the view point."""
    if not points.is_cuda:
        raise L.RoitrError("roitr_amd.prep needs ROCm device tensors (no CPU fallback)")
    points, normals = points.contiguous().float(), normals.contiguous().float()
    out = torch.empty_like(normals)
    L.check(L.lib().roitr_normal_redirect(int(points.shape[0]), L.ptr(points), L.ptr(normals), _vp(view_point), L.ptr(out), L.stream_ptr()),
            "normal_redirect")
    return out
