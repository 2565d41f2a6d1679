"""GPU: the engine mode operand_dtype = 'f32x3' (RoitrEngineConfig.operand_dtype = 2, round 6): fp32 everywhere, the plain linear layers
with K >= 256 multiply on the bf16 matrix cores by the three-way operand split of csrc/gemm_x3.hip.  The mode must satisfy the fp32
engine's own parity bars -- unchanged tolerances -- and its batch invariance bit for bit.  (The WHOLE -m gpu suite runs in this mode with
ROITR_TEST_OPERAND_DTYPE=f32x3, tests/gpu_util.py; round 6: 385 passed.  Kernel level: tests/test_stages_gpu.py::test_gemm_x3_*.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import oracle_forward  # noqa: E402
from corr_util import common_order_equal, compare_correspondences, to_numpy_corr  # noqa: E402
from test_correspondences_gpu import _check_against_oracle  # noqa: E402
from test_timed_shape_gpu import assert_bitwise  # noqa: E402

from roitr_amd.harness import build_model, pair_to_device  # noqa: E402
from roitr_amd.synthetic import make_pair  # noqa: E402


def test_f32x3_reference_golden_end_to_end():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pair_sel_n1024.npz"))
    model = build_model("3DMatch", operand_dtype="f32x3", weights="selective")
    pair = {k[3:]: g[k] for k in g.files if k.startswith("in.")}
    with torch.no_grad():
        out = model.forward(**pair_to_device(pair))
    for k in ("src_nodes", "tgt_nodes"):
        assert np.array_equal(out[k].cpu().numpy(), g["out." + k])
    for k in ("src_node_feats", "tgt_node_feats"):
        assert np.abs(out[k].cpu().numpy() - g["out." + k]).max() < 1e-4
    for k in ("src_point_feats", "tgt_point_feats"):
        assert np.abs(out[k].cpu().numpy()[::4] - g[f"out.{k}.every4"]).max() < 1e-4
    assert np.array_equal(out["tgt_node_corr_indices"].cpu().numpy(), g["out.tgt_node_corr_indices"])
    assert np.array_equal(out["src_node_corr_indices"].cpu().numpy(), g["out.src_node_corr_indices"])
    got = to_numpy_corr(out)
    want = {k: g["out." + k] for k in ("tgt_corr_points", "src_corr_points", "corr_scores")}
    frac, err, _ = compare_correspondences(got, want)
    assert frac >= 0.995 and err < 1e-4, (frac, err)
    assert common_order_equal(got, want)


def test_f32x3_matches_the_oracle_at_5000_and_is_batch_invariant():
    model = build_model("3DMatch", operand_dtype="f32x3", weights="selective")
    pair, ref = oracle_forward("3DMatch", 5000, 2, 1)
    pairs = [pair_to_device(pair)] + [pair_to_device(make_pair(n, config=2, pair_index=20 + i, normals="field")) for i, n in enumerate((3000, 5000, 1500, 4096, 2048, 5000, 1024))]
    with torch.no_grad():
        together = model.forward_batch(pairs)
        alone = [model.forward_batch([p])[0] for p in pairs]
    for i, (a, b) in enumerate(zip(together, alone)):
        assert_bitwise(a, b, f"f32x3: pair {i} in a batch of 8 vs alone")
    ir_g, ir_o = _check_against_oracle(together[0], ref, pair, coarse_exact=False)
    assert abs(ir_g - ir_o) <= 1e-3
