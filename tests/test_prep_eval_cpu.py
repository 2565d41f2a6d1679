"""SURVEY.md 8f rows on CPU: the oracles for normal_redirect / the evaluators against vectors captured from the imported
reference (tests/golden/make_golden.py --prep-eval), and the PCA-normal restatement against first principles."""
import os

import numpy as np

from oracle import eval_ref, prep_ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "prep_eval.npz"))


def test_normal_redirect_oracle_matches_reference():
    for i in range(3):
        out = prep_ref.normal_redirect(G["redirect.points"], G["redirect.normals"], G[f"redirect.view{i}"]).astype(np.float32)
        assert np.array_equal(out, G[f"redirect.out{i}"])


def test_evaluator_oracle_matches_reference():
    for case in range(3):
        p = f"eval{case}."
        ir = eval_ref.inlier_ratio(G[p + "src"], G[p + "tgt"], G[p + "rot"], G[p + "trans"], 0.1)
        assert abs(ir - float(G[p + "IR"])) < 1e-7
        if (p + "IR_bu") in G:
            assert abs(ir - float(G[p + "IR_bu"])) < 1e-7
        nt, ns = G[p + "n_nodes"]
        pir = eval_ref.coarse_precision(int(nt), int(ns), G[p + "gt_idx"], G[p + "gt_ov"], G[p + "tgt_corr"], G[p + "src_corr"], 0.0)
        assert abs(pir - float(G[p + "PIR"])) < 1e-7


def test_estimate_normals_oracle_on_planes():
    """PCA normal of points sampled on a plane = the plane normal; a thin slab gives it within the slab's aspect ratio."""
    rng = np.random.default_rng(3)
    n0 = np.array([1.0, 2.0, -0.5]); n0 /= np.linalg.norm(n0)
    u = np.cross(n0, [0, 0, 1.0]); u /= np.linalg.norm(u)
    v = np.cross(n0, u)
    ab = rng.uniform(-1, 1, (500, 2))
    pts = (ab[:, :1] * u + ab[:, 1:] * v + rng.normal(size=(500, 1)) * 1e-4 * n0).astype(np.float32)
    nrm, gap = prep_ref.estimate_normals(pts, 33)
    assert np.all(np.abs(nrm @ n0) > 1 - 1e-4)
    assert np.all(gap > 0.1)
    red = prep_ref.normal_redirect(pts.astype(np.float64), nrm, 5.0 * n0)
    assert np.all(red @ n0 > 0)
