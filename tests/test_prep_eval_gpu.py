"""SURVEY.md 8f rows on the GPU: normal estimation / redirect and the IR / PIR evaluators through the C ABI, against the
golden vectors captured from the reference and the CPU oracles."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "prep_eval.npz"))


def _dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dt is None else t.to(dt)


def test_normal_redirect_matches_reference_vectors():
    from roitr_amd import prep
    for i in range(3):
        out = prep.normal_redirect(_dev(G["redirect.points"]), _dev(G["redirect.normals"]), G[f"redirect.view{i}"])
        assert np.array_equal(out.cpu().numpy(), G[f"redirect.out{i}"])


@pytest.mark.parametrize("sizes", [(3000, 2500), (500, 300, 40), (5000,)])
def test_estimate_normals_matches_oracle(sizes):
    """Oriented PCA normals of several concatenated clouds (grid and brute-force kNN paths) against the float64 oracle;
    points whose two smallest eigenvalues nearly coincide (direction ill-conditioned) are compared with a looser bound."""
    from oracle import prep_ref
    from roitr_amd import prep
    rng = np.random.default_rng(11 + len(sizes))
    clouds = []
    for n in sizes:   # noisy curved surface patches: well-defined normals almost everywhere
        ab = rng.uniform(-1, 1, (n, 2))
        z = 0.3 * np.sin(2.0 * ab[:, 0]) * np.cos(1.5 * ab[:, 1]) + rng.normal(size=n) * 0.004
        clouds.append(np.stack([ab[:, 0] + 1.0, ab[:, 1] + 1.0, z + 1.0], 1).astype(np.float32))
    xyz = np.concatenate(clouds)
    off = np.cumsum(sizes).astype(np.int32)
    vp = (0.0, 0.0, 0.0)
    got = prep.estimate_normals(_dev(xyz), _dev(off), knn=33, view_point=vp).cpu().numpy().astype(np.float64)
    s = 0
    for c, n in zip(clouds, sizes):
        ref, gap = prep_ref.estimate_normals(c, 33)
        ref = prep_ref.normal_redirect(c.astype(np.float64), ref, np.asarray(vp))
        g = got[s:s + n]
        assert np.allclose(np.linalg.norm(g, axis=1), 1.0, atol=1e-6)
        dots = (g * ref).sum(1)
        good = gap > 1e-2
        # the sign is only comparable where the view vector is not (numerically) orthogonal to the normal
        side = np.abs(((np.asarray(vp) - c.astype(np.float64)) * ref).sum(1)) > 1e-6
        assert good.mean() > 0.95
        assert np.all(dots[good & side] > 1 - 1e-6), float(dots[good & side].min())
        assert np.all(np.abs(dots[~good]) > 1 - 1e-2)
        s += n


def test_estimate_normals_tiny_cloud_and_unoriented():
    from roitr_amd import prep
    pts = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0.0], [5, 5, 5], [6, 5, 5]], device="cuda")
    off = torch.tensor([4, 6], dtype=torch.int32, device="cuda")
    n = prep.estimate_normals(pts, off, knn=33, view_point=(0.5, 0.5, 3.0)).cpu().numpy()
    assert np.allclose(n[:4], [[0, 0, 1]] * 4, atol=1e-6)     # planar square, flipped towards +z
    assert np.allclose(np.abs(n[4:]), [[0, 0, 1]] * 2)        # 2 points: Open3D's (0, 0, 1) default
    raw = prep.estimate_normals(pts, off, knn=33, view_point=None).cpu().numpy()
    assert np.allclose(np.abs(raw[:4]), [[0, 0, 1]] * 4, atol=1e-6)


def test_evaluator_matches_reference_vectors():
    from roitr_amd.evaluate import Evaluator, get_inlier_ratio_correspondence
    ev = Evaluator(dict(eval_acceptance_overlap=0.0, eval_acceptance_radius=0.1))
    for case in range(3):
        p = f"eval{case}."
        od = dict(tgt_node_corr_indices=_dev(G[p + "tgt_corr"]).long(), src_node_corr_indices=_dev(G[p + "src_corr"]).long(),
                  gt_node_corr_indices=_dev(G[p + "gt_idx"]).long(), gt_node_corr_overlaps=_dev(G[p + "gt_ov"]),
                  tgt_corr_points=_dev(G[p + "tgt"]), src_corr_points=_dev(G[p + "src"]))
        dd = dict(rot=_dev(G[p + "rot"])[None], trans=_dev(G[p + "trans"])[None])
        res = ev(od, dd)
        assert abs(float(res["PIR"]) - float(G[p + "PIR"])) < 1e-7
        assert abs(float(res["IR"]) - float(G[p + "IR"])) < 1e-7
        if (p + "IR_bu") in G:
            ir = get_inlier_ratio_correspondence(_dev(G[p + "src"]), _dev(G[p + "tgt"]), _dev(G[p + "rot"]), _dev(G[p + "trans"]), 0.1)
            assert abs(float(ir) - float(G[p + "IR_bu"])) < 1e-7


def test_evaluate_batch_equals_per_pair_oracle():
    """IR / PIR of a whole engine batch (two launches) against the numpy oracle applied to every pair's outputs."""
    from oracle import eval_ref
    from roitr_amd.evaluate import Evaluator
    from roitr_amd.synthetic import make_pair
    from tests.gpu_util import build_model, pair_to_device
    model = build_model("3DMatch")
    pairs = [pair_to_device(make_pair(n, config=2, pair_index=i)) for i, n in enumerate((1024, 1500, 1024))]
    ev = Evaluator(dict(eval_acceptance_overlap=0.0, eval_acceptance_radius=0.1))
    with torch.no_grad():
        h = model.launch_batch(pairs, want_gt=True)
        ir, pir, n_fine, n_coarse = ev.evaluate_batch(h)
        res = model.finish_batch(h)
    for b, (r, p) in enumerate(zip(res, pairs)):
        c = lambda t: t.detach().cpu().numpy()
        ir_ref = eval_ref.inlier_ratio(c(r["src_corr_points"]), c(r["tgt_corr_points"]), c(p["rot"]).reshape(3, 3), c(p["trans"]).reshape(3), 0.1)
        assert abs(float(ir[b]) - ir_ref) < 1e-6 and int(n_fine[b]) == r["corr_scores"].shape[0]
        pir_ref = eval_ref.coarse_precision(r["tgt_nodes"].shape[0], r["src_nodes"].shape[0], c(r["gt_node_corr_indices"]),
                                            c(r["gt_node_corr_overlaps"]), c(r["tgt_node_corr_indices"]), c(r["src_node_corr_indices"]), 0.0)
        assert abs(float(pir[b]) - pir_ref) < 1e-6


def test_tester_evaluates_and_estimates_normals(tmp_path):
    """The test loop with on-device normal estimation in front and the PIR / IR evaluators behind the model: the
    metrics equal the per-pair reference formulas applied to the saved result files' inputs."""
    from oracle import eval_ref
    from roitr_amd.config import test_config
    from roitr_amd.tester import SyntheticPairs, Tester
    from tests.gpu_util import build_model
    model = build_model("3DMatch")
    data = SyntheticPairs(3, 1024, config=1)
    t = Tester(test_config("3DMatch"), model, data, str(tmp_path), pairs_per_forward=2, evaluate=True, estimate_normals=True)
    t.test()
    assert t.metrics["pairs"] == 3 and 0.0 <= t.metrics["IR"] <= 1.0 and 0.0 <= t.metrics["PIR"] <= 1.0
    irs = []
    for i in range(3):
        d = torch.load(tmp_path / "3DMatch" / f"{i}.pth")
        irs.append(eval_ref.inlier_ratio(d["src_corr_pts"].numpy(), d["tgt_corr_pts"].numpy(), d["rot"].numpy().reshape(3, 3),
                                         d["trans"].numpy().reshape(3), 0.1))
    assert abs(np.mean(irs) - t.metrics["IR"]) < 1e-6
