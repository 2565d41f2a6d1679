"""Error behaviour of the product path on the device: bad inputs end in a RoitrError that says what was wrong -- never in a silent
CPU fallback, a device fault, or a wrong result (the reference raises from torch / its CUDA extension at the same places)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(n, i=0):
    from gpu_util import pair_to_device
    from roitr_amd.synthetic import make_pair
    return pair_to_device(make_pair(n, config=2, pair_index=i))


def test_cloud_below_four_superpoints_is_refused():
    """model/model.py:59-62 leaves n // 64 superpoints; the engine needs at least 4 (kNN of the coarsest level)."""
    from gpu_util import build_model
    from roitr_amd import _lib as L
    model = build_model("3DMatch")
    with pytest.raises(L.RoitrError, match="cloud too small"):
        model.forward_batch([_pair(200)])
    # the engine is still usable afterwards
    out = model.forward_batch([_pair(1024, 1)])
    assert out[0]["src_node_feats"].shape[0] == 16


def test_host_tensors_are_refused_before_anything_is_launched():
    from gpu_util import build_model
    from roitr_amd import _lib as L
    from roitr_amd import ops, pointops as P
    model = build_model("3DMatch")
    pair = {k: v.cpu() for k, v in _pair(1024).items()}
    with pytest.raises(L.RoitrError, match="device tensors"):
        model.forward_batch([pair])
    x = torch.randn(64, 64)
    with pytest.raises(L.RoitrError, match="device tensors"):
        ops.linear(x, x)
    with pytest.raises(L.RoitrError):
        P.furthestsampling(torch.zeros(8, 3), torch.tensor([8], dtype=torch.int32), torch.tensor([2], dtype=torch.int32))


def test_unsupported_shapes_name_themselves():
    from roitr_amd import _lib as L
    from roitr_amd import ops
    M, N_in = 40, 200
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g).cuda()
    grp = torch.randint(0, N_in, (M, 16), generator=g).to(torch.int32).cuda()
    # in_dim 96 is not a TransitionDown width of the model
    with pytest.raises(L.RoitrError, match="in_dim"):
        ops.local_attention_fold(r(N_in, 96), r(M, 128), r(M, 4, 96), grp, r(M, 16, 4), r(128, 4), r(128, 4), r(128))
    with pytest.raises(L.RoitrError, match="16 neighbours"):
        ops.local_attention_fold(r(N_in, 64), r(M, 128), r(M, 4, 64), grp[:, :8].contiguous(), r(M, 8, 4), r(128, 4), r(128, 4), r(128))
    # a GEMM with a null operand
    gm = ops._Gemm(8, 64, 64, L.ptr(None), L.ptr(None), 64, L.ptr(None), 0, L.ptr(r(64, 64)), 64, L.ptr(None), 0, L.ptr(None), 1.0, 0,
                   L.ptr(r(8, 64)), 64, 1, 0, 0, 0, 0, 0, 0, L.ptr(None), 0, 0)
    with pytest.raises(L.RoitrError, match="null operand"):
        L.check(L.lib().roitr_gemm(ctypes.byref(gm), L.stream_ptr()), "gemm")
    # the interpolation addend exists in the fused LayerNorm epilogue only
    gm = ops._Gemm(8, 64, 64, L.ptr(r(8, 64)), L.ptr(None), 64, L.ptr(None), 0, L.ptr(r(64, 64)), 64, L.ptr(None), 0, L.ptr(None), 1.0, 0,
                   L.ptr(r(8, 64)), 64, 1, 0, 0, 0, 0, 0, 0, L.ptr(None), 0, 0)
    gm.ip_feat, gm.ip_idx, gm.ip_dist2 = L.ptr(r(4, 64)), L.ptr(torch.zeros(8, 3, dtype=torch.int32).cuda()), L.ptr(r(8, 3).abs())
    with pytest.raises(L.RoitrError, match="LayerNorm epilogue"):
        L.check(L.lib().roitr_gemm(ctypes.byref(gm), L.stream_ptr()), "gemm")
