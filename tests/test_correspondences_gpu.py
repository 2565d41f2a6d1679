"""GPU: END-TO-END VALUES of the forward -- the correspondences the tester saves (lib/tester.py:59-63) -- not only counts.

With the plain closed-form weights no fine-matching score clears the 0.05 confidence threshold at N = 1024 (the golden has 0
correspondences) and every 4DMatch node pair passes the 0.75 similarity threshold; the 'selective' weight variant
(roitr_amd/weights.py) on pairs with field normals makes both stages discriminate:
  * tests/golden/pair_sel_n1024.npz (captured from the reference): 4 740 correspondences, compared as a multiset of
    (tgt point, src point) with their scores, in torch.nonzero order, plus matching_scores and the inlier ratio;
  * full sizes against the CPU oracle (pinned to the reference by tests/test_oracle_cpu.py): 3DMatch N = 5000, 4DMatch N = 8000
    (a few percent of the 125^2 node pairs under the threshold: the coarse comparison can fail);
  * BASELINE config 3 as written: make_pair(config=3) -- the test-time SO(3) rotation of dataset/tdmatch.py:99-112 -- 8 pairs in
    ONE engine batch, per-pair correspondences and the mean Inlier Ratio (lib/loss.py:195-206) against the oracle: |dIR| <= 0.001.
Tolerances: indices / partitions identical; >= 98 % of the correspondences common (entries on a discrete boundary -- the
confidence threshold, a top-k tie -- may fall on either side under fp32 reordering), scores of common entries within 1e-4,
optimal-transport entries within 2e-4 (relative above magnitude 1).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import oracle_forward  # noqa: E402
from corr_util import (assert_descriptors_close, common_order_equal, compare_correspondences, inlier_ratio,  # noqa: E402
                       matching_scores_error, to_numpy_corr)
from gpu_util import build_model, pair_to_device  # noqa: E402


def _np(out, keys):
    return {k: out[k].detach().cpu().numpy() for k in keys}


def test_selective_golden_end_to_end():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pair_sel_n1024.npz"))
    model = build_model("3DMatch", weights="selective")
    pair = {k[3:]: g[k] for k in g.files if k.startswith("in.")}
    with torch.no_grad():
        out = model.forward(**pair_to_device(pair))
    for k in ("src_nodes", "tgt_nodes"):
        assert np.array_equal(out[k].cpu().numpy(), g["out." + k])
    for k in ("src_node_feats", "tgt_node_feats"):
        assert np.abs(out[k].cpu().numpy() - g["out." + k]).max() < 1e-4
    for k in ("src_point_feats", "tgt_point_feats"):       # |values| up to 11 (fine_proj gain 4): relative AND absolute bound, measured values printed
        assert_descriptors_close(out[k].cpu().numpy()[::4], g[f"out.{k}.every4"], k)
    assert np.array_equal(out["tgt_node_corr_indices"].cpu().numpy(), g["out.tgt_node_corr_indices"])
    assert np.array_equal(out["src_node_corr_indices"].cpu().numpy(), g["out.src_node_corr_indices"])
    ms, ref = out["matching_scores"].cpu().numpy()[::4], g["out.matching_scores.every4"]
    tm = np.concatenate([out["tgt_node_corr_knn_masks"].cpu().numpy()[::4], np.ones((ref.shape[0], 1), bool)], 1)
    sm = np.concatenate([out["src_node_corr_knn_masks"].cpu().numpy()[::4], np.ones((ref.shape[0], 1), bool)], 1)
    valid = tm[:, :, None] & sm[:, None, :]
    assert (np.abs(ms - ref) / np.maximum(1.0, np.abs(ref)))[valid].max() < 2e-4
    got = to_numpy_corr(out)
    want = {k: g["out." + k] for k in ("tgt_corr_points", "src_corr_points", "corr_scores")}
    assert want["corr_scores"].shape[0] > 1000            # a real set (4 740 with the committed golden), not the `0 == 0` of the plain weights
    frac, err, common = compare_correspondences(got, want)
    assert frac >= 0.995, (frac, got["corr_scores"].shape)
    assert err < 1e-4, err
    assert common_order_equal(got, want)                    # torch.nonzero order of modules.py:316-322
    assert abs(inlier_ratio(got, pair["rot"], pair["trans"]) - inlier_ratio(want, pair["rot"], pair["trans"])) <= 1e-3


def _check_against_oracle(out, ref, pair, coarse_exact):
    for k in ("src_nodes", "tgt_nodes"):
        assert np.array_equal(out[k].cpu().numpy(), ref[k]), k
    for side in ("src", "tgt"):
        assert np.array_equal(out[f"_{side}_node_knn_indices"].cpu().numpy(), ref[f"_{side}_node_knn_indices"]), side
    for k in ("src_node_feats", "tgt_node_feats", "src_point_feats", "tgt_point_feats"):
        assert_descriptors_close(out[k].cpu().numpy(), ref[k], k)
    got_c = set(zip(out["tgt_node_corr_indices"].tolist(), out["src_node_corr_indices"].tolist()))
    want_c = set(zip(ref["tgt_node_corr_indices"].tolist(), ref["src_node_corr_indices"].tolist()))
    if coarse_exact:
        assert got_c == want_c
    else:
        assert len(got_c & want_c) >= 0.98 * len(want_c), (len(got_c & want_c), len(want_c))
    o = _np(out, ("tgt_node_corr_indices", "src_node_corr_indices", "matching_scores"))
    worst, n = matching_scores_error(o, ref)
    assert n >= 0.98 * len(want_c) and worst < 2e-4, (worst, n)
    got = to_numpy_corr(out)
    frac, err, common = compare_correspondences(got, ref)
    assert ref["corr_scores"].shape[0] > 100
    assert frac >= 0.98, (frac, got["corr_scores"].shape, ref["corr_scores"].shape)
    assert err < 1e-4, err
    return inlier_ratio(got, pair["rot"], pair["trans"]), inlier_ratio(ref, pair["rot"], pair["trans"])


def test_3dmatch_correspondences_at_5000_match_oracle():
    pair, ref = oracle_forward("3DMatch", 5000, 2, 1)
    model = build_model("3DMatch", weights="selective")
    with torch.no_grad():
        out = model.forward(**pair_to_device(pair))
    ir_g, ir_o = _check_against_oracle(out, ref, pair, coarse_exact=False)
    assert abs(ir_g - ir_o) <= 1e-3


def test_3dmatch_correspondences_on_a_scan_like_pair_match_oracle():
    """Round 4: piecewise-planar, room-like clouds with sensor noise (synthetic.surface_points) -- the geometry of real 3DMatch
    fragments: grid cells are mostly empty, the occupied ones dense, kNN ties and plane-degenerate PPFs are frequent.  Same checks
    as on the uniform clouds: nodes, partition, descriptors, coarse and fine correspondences against the CPU oracle."""
    pair, ref = oracle_forward("3DMatch", 5000, 2, 3, cloud="surface")
    model = build_model("3DMatch", weights="selective")
    with torch.no_grad():
        out = model.forward(**pair_to_device(pair))
    ir_g, ir_o = _check_against_oracle(out, ref, pair, coarse_exact=False)
    assert abs(ir_g - ir_o) <= 1e-3


def test_4dmatch_correspondences_at_8000_match_oracle():
    """BASELINE config 4 sizes in fp32.  The coarse stage is selective here: a few percent of the 15 625 node pairs lie under the
    0.75 threshold (with the plain weights it was all of them), so the set comparison is a real one."""
    pair, ref = oracle_forward("4DMatch", 8000, 4, 2)
    n_sel = len(ref["tgt_node_corr_indices"])
    assert 128 < n_sel < 0.5 * 125 * 125, n_sel             # threshold branch, not every pair
    model = build_model("4DMatch", weights="selective")
    with torch.no_grad():
        out = model.forward(**pair_to_device(pair))
    ir_g, ir_o = _check_against_oracle(out, ref, pair, coarse_exact=False)
    assert abs(ir_g - ir_o) <= 1e-3


def test_config3_rotated_batch_of_8_and_inlier_ratio():
    """BASELINE config 3: 3DLoMatch-rotated pairs (make_pair(config=3)), 8 pairs in one engine call; per pair the forward matches
    the oracle's single-pair forward, and the mean inlier ratio over the 8 pairs agrees to 0.1 pp (north star)."""
    items = [oracle_forward("3DLoMatch", 5000, 3, i) for i in range(8)]
    model = build_model("3DLoMatch", weights="selective")
    with torch.no_grad():
        outs = model.forward_batch([pair_to_device(p) for p, _ in items])
    irs_g, irs_o = [], []
    for out, (pair, ref) in zip(outs, items):
        a, b = _check_against_oracle(out, ref, pair, coarse_exact=False)
        irs_g.append(a)
        irs_o.append(b)
    assert abs(np.mean(irs_g) - np.mean(irs_o)) <= 1e-3, (irs_g, irs_o)
    assert max(abs(a - b) for a, b in zip(irs_g, irs_o)) <= 2e-3
