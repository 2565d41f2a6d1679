"""Python operator API of the reference's `pointops`, backed by libroitr_hip.so.

Mirrors cpp_wrappers/pointops/functions/pointops.py (reference) name for name: same arguments,
same return conventions (int32 indices from the native calls, euclidean distances from knnquery,
int64 group indices from queryandgroup), same contiguity asserts.  Differences, all additive:
kernels run on torch's CURRENT stream (the reference used the legacy default stream), calls
return an error instead of failing silently, and `knn_ppf` exposes the fused
queryandgroup+calc_ppf_gpu pass the engine uses.

Tensors must live on a ROCm device; there is no CPU path here.
"""
import ctypes

import torch

from . import _lib as L

# clouds at or below this many reference points are scanned brute force (index order)
GRID_MIN_POINTS = 768


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.RoitrError("roitr_amd.pointops needs ROCm device tensors (no CPU fallback)")


def _i32(t):
    return t if t.dtype == torch.int32 else t.to(torch.int32)


def furthestsampling(xyz, offset, new_offset):
    """pointops.py:10-27.  xyz (n,3) f32, offset/new_offset (b,) i32 cumulative -> idx (m,) i32."""
    assert xyz.is_contiguous()
    _need_gpu(xyz, offset, new_offset)
    offset, new_offset = _i32(offset).contiguous(), _i32(new_offset).contiguous()
    n, b = xyz.shape[0], offset.shape[0]
    off_h = offset.tolist()
    n_max = off_h[0]
    for i in range(1, b):
        n_max = max(off_h[i] - off_h[i - 1], n_max)
    m = int(new_offset[b - 1].item())
    idx = torch.zeros(m, dtype=torch.int32, device=xyz.device)
    tmp = torch.full((n,), 1e10, dtype=torch.float32, device=xyz.device)
    L.check(L.lib().roitr_furthestsampling(b, int(n_max), L.ptr(xyz), L.ptr(offset), L.ptr(new_offset), L.ptr(tmp),
                                          L.ptr(idx), L.stream_ptr()), "furthestsampling")
    return idx


def _knn(nsample, xyz, new_xyz, offset, new_offset, want_idx=True, want_dist=True, want_group=False,
         ref_normals=None, query_normals=None, use_grid=None):
    if new_xyz is None:
        new_xyz = xyz
    assert xyz.is_contiguous() and new_xyz.is_contiguous()
    _need_gpu(xyz, new_xyz, offset, new_offset)
    offset, new_offset = _i32(offset).contiguous(), _i32(new_offset).contiguous()
    n, m, b = xyz.shape[0], new_xyz.shape[0], offset.shape[0]
    dev = xyz.device
    # every element of the outputs is written by the query kernels (the engine hands them uninitialised arena memory too)
    idx = torch.empty((m, nsample), dtype=torch.int32, device=dev) if want_idx else None
    d2 = torch.empty((m, nsample), dtype=torch.float32, device=dev) if want_dist else None
    grp = torch.empty((m, nsample - 1), dtype=torch.int32, device=dev) if want_group else None
    ppf = None
    if ref_normals is not None:
        assert ref_normals.is_contiguous() and query_normals.is_contiguous()
        ppf = torch.empty((m, nsample - 1, 4), dtype=torch.float32, device=dev)
    if use_grid is None:
        use_grid = n > GRID_MIN_POINTS * b
    lib = L.lib()
    ws = torch.empty(lib.roitr_knn_workspace_bytes(b, n, m), dtype=torch.uint8, device=dev)
    st = L.stream_ptr()
    if use_grid:
        # ring-1 neighbourhood (27 cells) should hold the k+1 neighbours with room for the guarantee radius; the
        # workgroup-per-cell kernel (k + 2 >= 35, self queries) measured best at 0.4 (k + 2) points per cell
        need = nsample + 1
        rho = ctypes.c_float(max(6.0, need * 0.4 if need >= 35 else need / 3.0))
        L.check(lib.roitr_knn_build_grid_ex(b, n, m, L.ptr(xyz), L.ptr(offset), L.ptr(ws), rho, st), "knn_build_grid")
    L.check(lib.roitr_knnquery_ex(b, n, m, int(nsample), L.ptr(xyz), L.ptr(new_xyz), L.ptr(offset), L.ptr(new_offset),
                                  L.ptr(idx), L.ptr(d2), L.ptr(grp), L.ptr(ppf), L.ptr(ref_normals), L.ptr(query_normals),
                                  1 if use_grid else 0, m, L.ptr(ws), st), "knnquery")
    return idx, d2, grp, ppf


def knnquery(nsample, xyz, new_xyz, offset, new_offset, use_grid=None):
    """pointops.py:30-45 -> (idx (m,nsample) i32, euclidean dist (m,nsample) f32)."""
    idx, d2, _, _ = _knn(nsample, xyz, new_xyz, offset, new_offset, use_grid=use_grid)
    return idx, torch.sqrt(d2)


def knnquery_raw(nsample, xyz, new_xyz, offset, new_offset, use_grid=None):
    """The native call alone: (idx i32, SQUARED distances)."""
    idx, d2, _, _ = _knn(nsample, xyz, new_xyz, offset, new_offset, use_grid=use_grid)
    return idx, d2


def knn_ppf(nsample, xyz, new_xyz, normals, new_normals, offset, new_offset, use_grid=None):
    """Fused queryandgroup(nsample, ..., return_idx=True) + calc_ppf_gpu (model/model.py:75-77):
    -> (group_idx (m,nsample) i32, ppf (m,nsample,4) f32)."""
    _, _, grp, ppf = _knn(nsample + 1, xyz, new_xyz, offset, new_offset, want_idx=False, want_dist=False,
                          want_group=True, ref_normals=normals, query_normals=new_normals, use_grid=use_grid)
    return grp, ppf


def queryandgroup(nsample, xyz, new_xyz, feat, idx, offset, new_offset, return_idx=False, use_xyz=True):
    """pointops.py:79-104."""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    if new_xyz is None:
        new_xyz = xyz
    if idx is None:
        _, _, grp, _ = _knn(nsample + 1, xyz, new_xyz, offset, new_offset, want_idx=False, want_dist=False, want_group=True)
        idx = grp.long()
    if return_idx:
        return idx
    m, c = new_xyz.shape[0], feat.shape[1]
    grouped_xyz = xyz[idx.view(-1).long(), :].view(m, nsample, 3)
    grouped_xyz -= new_xyz.unsqueeze(1)
    grouped_feat = feat[idx.view(-1).long(), :].view(m, nsample, c)
    if use_xyz:
        return torch.cat((grouped_xyz, grouped_feat), -1)
    return grouped_feat


def _status_call(fn, *args):
    L.check(fn(*args, L.stream_ptr()), fn.__name__)


class Grouping(torch.autograd.Function):
    """pointops.py:48-76."""

    @staticmethod
    def forward(ctx, input, idx):
        assert input.is_contiguous() and idx.is_contiguous()
        _need_gpu(input, idx)
        idx = _i32(idx)
        m, nsample, n, c = idx.shape[0], idx.shape[1], input.shape[0], input.shape[1]
        output = torch.empty((m, nsample, c), dtype=torch.float32, device=input.device)
        _status_call(L.lib().roitr_grouping_forward, m, nsample, c, L.ptr(input), L.ptr(idx), L.ptr(output))
        ctx.n = n
        ctx.save_for_backward(idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        idx, = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        m, nsample, c = grad_output.shape
        grad_input = torch.zeros((ctx.n, c), dtype=torch.float32, device=grad_output.device)
        _status_call(L.lib().roitr_grouping_backward, m, nsample, c, L.ptr(grad_output), L.ptr(idx), L.ptr(grad_input))
        return grad_input, None


grouping = Grouping.apply


class Subtraction(torch.autograd.Function):
    """pointops.py:107-134."""

    @staticmethod
    def forward(ctx, input1, input2, idx):
        assert input1.is_contiguous() and input2.is_contiguous()
        _need_gpu(input1, input2, idx)
        idx = _i32(idx).contiguous()
        n, c = input1.shape
        nsample = idx.shape[-1]
        output = torch.zeros((n, nsample, c), dtype=torch.float32, device=input1.device)
        _status_call(L.lib().roitr_subtraction_forward, n, nsample, c, L.ptr(input1), L.ptr(input2), L.ptr(idx), L.ptr(output))
        ctx.save_for_backward(idx)
        ctx.n2 = input2.shape[0]
        return output

    @staticmethod
    def backward(ctx, grad_output):
        idx, = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, nsample, c = grad_output.shape
        g1 = torch.zeros((n, c), dtype=torch.float32, device=grad_output.device)
        g2 = torch.zeros((ctx.n2, c), dtype=torch.float32, device=grad_output.device)
        _status_call(L.lib().roitr_subtraction_backward, n, nsample, c, L.ptr(idx), L.ptr(grad_output), L.ptr(g1), L.ptr(g2))
        return g1, g2, None


subtraction = Subtraction.apply


class Aggregation(torch.autograd.Function):
    """pointops.py:137-165."""

    @staticmethod
    def forward(ctx, input, position, weight, idx):
        assert input.is_contiguous() and position.is_contiguous() and weight.is_contiguous()
        _need_gpu(input, position, weight, idx)
        idx = _i32(idx).contiguous()
        n, nsample, c = position.shape
        w_c = weight.shape[-1]
        output = torch.zeros((n, c), dtype=torch.float32, device=input.device)
        _status_call(L.lib().roitr_aggregation_forward, n, nsample, c, w_c, L.ptr(input), L.ptr(position), L.ptr(weight),
                     L.ptr(idx), L.ptr(output))
        ctx.save_for_backward(input, position, weight, idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input, position, weight, idx = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, nsample, c = position.shape
        w_c = weight.shape[-1]
        gi = torch.zeros_like(input)
        gp = torch.zeros_like(position)
        gw = torch.zeros_like(weight)
        _status_call(L.lib().roitr_aggregation_backward, n, nsample, c, w_c, L.ptr(input), L.ptr(position), L.ptr(weight),
                     L.ptr(idx), L.ptr(grad_output), L.ptr(gi), L.ptr(gp), L.ptr(gw))
        return gi, gp, gw, None


aggregation = Aggregation.apply


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """pointops.py:168-182 (the Python composite the model calls, model/model.py:116)."""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    idx, dist = knnquery(k, xyz, new_xyz, offset, new_offset)
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=1, keepdim=True)
    weight = dist_recip / norm
    new_feat = torch.zeros((new_xyz.shape[0], feat.shape[1]), dtype=torch.float32, device=feat.device)
    for i in range(k):
        new_feat += feat[idx[:, i].long(), :] * weight[:, i].unsqueeze(-1)
    return new_feat


class Interpolation(torch.autograd.Function):
    """pointops.py:185-218."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, input, offset, new_offset, k=3):
        assert xyz.is_contiguous() and new_xyz.is_contiguous() and input.is_contiguous()
        idx, dist = knnquery(k, xyz, new_xyz, offset, new_offset)
        dist_recip = 1.0 / (dist + 1e-8)
        norm = torch.sum(dist_recip, dim=1, keepdim=True)
        weight = (dist_recip / norm).contiguous()
        n, c, m = new_xyz.shape[0], input.shape[1], input.shape[0]
        output = torch.zeros((n, c), dtype=torch.float32, device=input.device)
        _status_call(L.lib().roitr_interpolation_forward, n, c, k, L.ptr(input), L.ptr(idx), L.ptr(weight), L.ptr(output))
        ctx.m, ctx.k = m, k
        ctx.save_for_backward(idx, weight)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        m, k = ctx.m, ctx.k
        idx, weight = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, c = grad_output.shape
        grad_input = torch.zeros((m, c), dtype=torch.float32, device=grad_output.device)
        _status_call(L.lib().roitr_interpolation_backward, n, c, k, L.ptr(grad_output), L.ptr(idx), L.ptr(weight), L.ptr(grad_input))
        return None, None, grad_input, None, None, None


interpolation2 = Interpolation.apply
