"""Operator-level mirrors of the reference's lib/utils.py hot-path helpers, backed by HIP kernels.

Each function cites the reference function it stands for.  ROCm device tensors only.
"""
import torch

from . import _lib as L


def _i32c(t):
    return t.to(torch.int32).contiguous()


def calc_ppf(points, point_normals, ref_points, ref_normals, group_idx):
    """lib/utils.py:358-389 calc_ppf_gpu(points, point_normals, ref_points[group_idx], ref_normals[group_idx]).

    The reference takes pre-gathered (m,k,3) patches; gathering is fused here, so the caller passes the
    un-gathered reference cloud and the (m,k) indices instead.  Returns (m,k,4) float32."""
    m, k = group_idx.shape
    out = torch.empty((m, k, 4), dtype=torch.float32, device=points.device)
    grp = _i32c(group_idx)
    L.check(L.lib().roitr_calc_ppf(m, k, L.ptr(points.contiguous()), L.ptr(point_normals.contiguous()),
                                   L.ptr(ref_points.contiguous()), L.ptr(ref_normals.contiguous()), L.ptr(grp), L.ptr(out),
                                   L.stream_ptr()), "calc_ppf")
    return out


def calc_ppf_gpu(points, point_normals, patches, patch_normals):
    """Signature-compatible with lib/utils.py:358 (pre-gathered patches (m,k,3))."""
    m, k, _ = patches.shape
    grp = torch.arange(m * k, dtype=torch.int32, device=points.device).view(m, k)
    return calc_ppf(points, point_normals, patches.reshape(-1, 3), patch_normals.reshape(-1, 3), grp)
