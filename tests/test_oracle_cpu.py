"""CPU: the oracle (oracle/pointops_ref.c + oracle/roitr_ref.py) against tensors captured from the reference.

These tests pin the oracle.  The native FPS/kNN goldens were produced by the reference's own Python glue
driving oracle/pointops_ref.c (the CUDA originals cannot run here), so for those the test checks glue +
determinism; everything downstream (PPF, attention, global transformer, matching) is pinned by the real
reference Python.
"""
import numpy as np
import pytest

from oracle import pointops_cpu as O
from oracle import roitr_ref as R

ATOL = 2e-5


def inputs(g):
    return {k[3:]: g[k] for k in g.files if k.startswith("in.")}


@pytest.fixture(scope="module")
def oracle_run(golden_pair):
    taps = {}
    out = R.forward(R.closed_form_state(), inputs(golden_pair), taps=taps, threads=4)
    return golden_pair, out, taps


def test_fps_and_knn_calls_reproduce(golden_pair):
    g = golden_pair
    p = g["in.raw_src_pcd"]
    n = p.shape[0]
    idx = O.furthestsampling(p, np.array([n], np.int32), np.array([n // 4], np.int32))
    assert np.array_equal(idx, g["fps.0"])
    o = np.array([n], np.int32)
    kidx, kd = O.knnquery(9, p, p, o, o)
    assert np.array_equal(kidx, g["knn.0.idx"])
    # torch-CPU sqrt (Sleef) is not always correctly rounded: 1-ulp slack on the euclidean distances only
    np.testing.assert_allclose(kd, g["knn.0.dist"], rtol=2e-7, atol=0)
    assert O.opt_n_threads(5000) == 1024 and O.opt_n_threads(312) == 256 and O.opt_n_threads(78) == 64


def test_fps_tie_rule_is_the_block_tournament():
    """Equal maxima: the reference's shared-memory tree keeps the lower SLOT, so the winner is decided by the
    bit-reversed thread id, not by the lowest index (sampling_cuda_kernel.cu:5-10,64-123)."""
    pts = np.zeros((8, 3), np.float32)
    pts[1:, 0] = 1.0  # seven points tie at distance 1 from point 0
    idx = O.furthestsampling(pts, np.array([8], np.int32), np.array([2], np.int32))
    assert idx.tolist() == [0, 4]  # bitrev3(4) = 1 is the smallest among tids 1..7


def test_knn_fill_when_cloud_smaller_than_k():
    pts = np.random.default_rng(0).random((5, 3)).astype(np.float32)
    o = np.array([5], np.int32)
    idx, d2 = O.knnquery_raw(8, pts, pts, o, o)
    assert (idx[:, 5:] == 0).all() and (d2[:, 5:] == np.float32(1e10)).all()


def test_ppf_stage(golden_stages):
    s = golden_stages
    out = R.calc_ppf(s["ppf.pts"], s["ppf.nrm"], s["ppf.patches"], s["ppf.pnrm"])
    np.testing.assert_allclose(out, s["ppf.out"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("stage", ["enc1.0", "enc1.1", "enc2.0", "enc2.2", "enc3.2", "enc4.2", "dec4.1", "dec3.1", "dec2.1", "dec1.1"])
def test_backbone_stages(oracle_run, stage):
    g, out, taps = oracle_run
    for j, tag in enumerate(("src", "tgt")):
        ref = g[f"feat.{stage}.{j}"]
        err = np.abs(taps[f"{tag}.{stage}"] - ref).max()
        assert err < ATOL, f"{tag}.{stage}: {err:.2e}"


def test_global_transformer_and_descriptors(oracle_run):
    g, out, taps = oracle_run
    for i in range(6):
        for j in range(2):
            err = np.abs(taps[f"geo.layer{i}"][j] - g[f"feat.geo.layer{i}.{j}"][0]).max()
            assert err < ATOL, f"geo.layer{i}.{j}: {err:.2e}"
    for k in ("src_node_feats", "tgt_node_feats", "src_point_feats", "tgt_point_feats"):
        assert np.abs(out[k] - g["out." + k]).max() < ATOL, k
    for k in ("src_nodes", "tgt_nodes"):
        assert np.array_equal(out[k], g["out." + k])


def test_geo_embedding_stage(golden_stages):
    s = golden_stages
    d, a = R.geo_embedding_indices(s["geo.points"][0])
    np.testing.assert_allclose(d, s["geo.d_idx"][0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(a, s["geo.a_idx"][0], rtol=0, atol=2e-6)
    W = R.Weights(R.closed_form_state())
    e = R.geo_embedding(W, "backbone.global_transformer.embedding", s["geo.points"][0], 256)
    np.testing.assert_allclose(e, s["geo.emb"][0], rtol=0, atol=ATOL)


def test_partition_and_matching(oracle_run, golden_stages):
    g, out, taps = oracle_run
    for side in ("src", "tgt"):
        assert np.array_equal(out[f"_{side}_node_knn_indices"], g[f"part.{side}.knn_indices"])
        assert np.array_equal(out[f"_{side}_node_knn_masks"], g[f"part.{side}.knn_masks"])
    assert sorted(zip(out["tgt_node_corr_indices"].tolist(), out["src_node_corr_indices"].tolist())) == \
        sorted(zip(g["out.tgt_node_corr_indices"].tolist(), g["out.src_node_corr_indices"].tolist()))
    s = golden_stages
    for tag in ("p0", "p1"):
        p2n, m, knn, km = R.point_to_node_partition(s[f"part.{tag}.points"], s[f"part.{tag}.nodes"], 64)
        assert np.array_equal(p2n, s[f"part.{tag}.point_to_node"])
        assert np.array_equal(knn, s[f"part.{tag}.knn_indices"]) and np.array_equal(km, s[f"part.{tag}.knn_masks"])
    ri, si, sc = R.coarse_matching(s["coarse.ref_f"], s["coarse.src_f"], s["coarse.ref_m"], s["coarse.src_m"], 256)
    assert np.array_equal(ri, s["coarse.ref_idx"]) and np.array_equal(si, s["coarse.src_idx"])
    np.testing.assert_allclose(sc, s["coarse.scores"], rtol=1e-4)


def test_ot_and_fine_stage(golden_stages):
    s = golden_stages
    ot = R.optimal_transport(s["ot.scores"], s["ot.row_masks"], s["ot.col_masks"], float(s["ot.alpha"]))
    rm = np.concatenate([s["ot.row_masks"], np.ones((8, 1), bool)], 1)
    cm = np.concatenate([s["ot.col_masks"], np.ones((8, 1), bool)], 1)
    valid = rm[:, :, None] & cm[:, None, :]
    assert np.abs(ot - s["ot.out"])[valid].max() < 1e-4
    for k, mutual in ((3, True), (2, True), (3, False)):
        tag = f"fine.k{k}.m{int(mutual)}"
        r, c, sc = R.fine_matching(s["fine.ref_pts"], s["fine.src_pts"], s["ot.row_masks"], s["ot.col_masks"],
                                   s["ot.out"][:, :-1, :-1], k, mutual)
        assert np.array_equal(r, s[tag + ".ref"]) and np.array_equal(c, s[tag + ".src"])
        np.testing.assert_allclose(sc, s[tag + ".scores"], rtol=1e-5)


def _ot_wide_check(ot, g, tol):
    B = g["scores"].shape[0]
    rm = np.concatenate([g["row_masks"], np.ones((B, 1), bool)], 1)
    cm = np.concatenate([g["col_masks"], np.ones((B, 1), bool)], 1)
    valid = rm[:, :, None] & cm[:, None, :]
    assert np.isfinite(ot[valid]).all()
    for b in range(B):
        err = (np.abs(ot[b] - g["out"][b]) / np.maximum(1.0, np.abs(g["out"][b])))[valid[b]].max()
        assert err < tol, (b, float(err))


def test_ot_wide_score_ranges():
    """Optimal transport on patches whose scores sit hundreds above / below the dustbin score or span +-100 (tests/golden/
    ot_wide.npz, captured from the reference's log-domain layer): the regime an exponential-domain Sinkhorn cannot hold in fp32."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ot_wide.npz"))
    ot = R.optimal_transport(g["scores"], g["row_masks"], g["col_masks"], float(g["alpha"]))
    _ot_wide_check(ot, g, 2e-5)


def test_adaptive_matching_stage(golden_stages):
    s = golden_stages
    for tag, mn in (("adaptive", 128), ("adaptive_nz", 32)):
        ia, ib, sc = R.adaptive_matching(s["coarse.ref_f"], s["coarse.src_f"], s["coarse.ref_m"], s["coarse.src_m"], mn, 0.75)
        assert np.array_equal(ia, s[f"{tag}.a_idx"]) and np.array_equal(ib, s[f"{tag}.b_idx"])
        np.testing.assert_allclose(sc, s[f"{tag}.scores"], rtol=1e-5)


def test_oracle_4dmatch_forward_matches_reference_golden():
    """The oracle in its 4DMatch configuration (factor 2 widths, AdaptiveSuperPointMatching, top-2 fine matching) against the
    end-to-end tensors captured from the reference (make_golden.py fdmatch_golden): pins the checker the full-size 4DMatch
    GPU tests (tests/test_fullsize_gpu.py) rely on."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pair_4dmatch_n1024.npz"))
    out = R.forward(R.closed_form_state(2, "selective"), inputs(g), cfg=dict(R.FDMATCH_CFG), threads=4)
    for k in ("src_nodes", "tgt_nodes"):
        assert np.array_equal(out[k], g["out." + k])
    for k in ("src_node_feats", "tgt_node_feats"):
        assert np.abs(out[k] - g["out." + k]).max() < ATOL
    for k in ("src_point_feats", "tgt_point_feats"):
        assert np.abs(out[k][::8] - g[f"out.{k}.every8"]).max() < ATOL
    assert np.array_equal(out["tgt_node_corr_indices"], g["out.tgt_node_corr_indices"])
    assert np.array_equal(out["src_node_corr_indices"], g["out.src_node_corr_indices"])
    ms, ref = out["matching_scores"][::8], g["out.matching_scores.every8"]
    tm = np.concatenate([out["tgt_node_corr_knn_masks"][::8], np.ones((ref.shape[0], 1), bool)], 1)
    sm = np.concatenate([out["src_node_corr_knn_masks"][::8], np.ones((ref.shape[0], 1), bool)], 1)
    valid = tm[:, :, None] & sm[:, None, :]
    big = np.maximum(1.0, np.abs(ref))   # log-space entries reach -200 with the selective weights: relative there
    assert (np.abs(ms - ref) / big)[valid].max() < 1e-4
    _check_correspondences(out, g)


def _check_correspondences(out, g, min_common=0.995):
    """End-to-end VALUES against the reference: the (tgt point, src point) multiset and the scores of the common entries; the
    inlier ratio (lib/loss.py:195-206) of both sets within 0.1 pp.  A handful of entries sit on a discrete boundary (the 0.05
    confidence threshold, a top-k tie) where numpy and torch-CPU rounding may fall on different sides."""
    from corr_util import compare_correspondences, inlier_ratio
    ref = {k: g["out." + k] for k in ("tgt_corr_points", "src_corr_points", "corr_scores")}
    assert ref["corr_scores"].shape[0] > 1000          # a non-trivial set: the golden is not the `0 == 0` of the plain weights
    frac, err, common = compare_correspondences(out, ref)
    assert frac >= min_common, (frac, out["corr_scores"].shape, ref["corr_scores"].shape)
    assert err < 1e-4, err
    ir_o, ir_r = inlier_ratio(out, g["in.rot"], g["in.trans"]), inlier_ratio(ref, g["in.rot"], g["in.trans"])
    assert abs(ir_o - ir_r) <= 1e-3, (ir_o, ir_r)


def test_oracle_selective_forward_matches_reference_golden():
    """3DMatch settings with the 'selective' weight variant on a pair with field normals (tests/golden/pair_sel_n1024.npz,
    captured from the reference): the golden whose forward ends in 4 740 correspondences -- compares what the tester saves."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pair_sel_n1024.npz"))
    taps = {}
    out = R.forward(R.closed_form_state(1, "selective"), inputs(g), taps=taps, threads=4)
    for k in ("src_nodes", "tgt_nodes"):
        assert np.array_equal(out[k], g["out." + k])
    for j, tag in enumerate(("src", "tgt")):
        for stage in ("enc4.2", "dec1.1"):
            assert np.abs(taps[f"{tag}.{stage}"] - g[f"feat.{stage}.{j}"]).max() < ATOL, (tag, stage)
    for k in ("src_node_feats", "tgt_node_feats"):
        assert np.abs(out[k] - g["out." + k]).max() < ATOL
    for k in ("src_point_feats", "tgt_point_feats"):   # |values| up to 11 (fine_proj gain 4)
        assert np.abs(out[k][::4] - g[f"out.{k}.every4"]).max() < 5e-5
    assert np.array_equal(out["tgt_node_corr_indices"], g["out.tgt_node_corr_indices"])
    assert np.array_equal(out["src_node_corr_indices"], g["out.src_node_corr_indices"])
    _check_correspondences(out, g)
