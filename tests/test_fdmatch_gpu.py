"""GPU: the 4DMatch configuration (factor 2 widths, AdaptiveSuperPointMatching, top-2 fine matching) against
tensors captured from the reference (tests/golden/pair_4dmatch_n1024.npz) and the stage known answers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import build_model, pair_to_device  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_adaptive_matching_stage(golden_stages):
    from roitr_amd import ops
    s = golden_stages
    for tag, mn in (("adaptive", 128), ("adaptive_nz", 32)):   # top-k branch / all-under-threshold branch
        ia, ib, sc = ops.adaptive_superpoint_matching(dev(s["coarse.ref_f"]), dev(s["coarse.src_f"]), dev(s["coarse.ref_m"]),
                                                      dev(s["coarse.src_m"]), mn, 0.75)
        assert np.array_equal(ia.cpu().numpy(), s[f"{tag}.a_idx"]) and np.array_equal(ib.cpu().numpy(), s[f"{tag}.b_idx"])
        np.testing.assert_allclose(sc.cpu().numpy(), s[f"{tag}.scores"], rtol=1e-5)


def test_fdmatch_forward():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pair_4dmatch_n1024.npz"))
    model = build_model("4DMatch", weights="selective")   # the golden was captured with the selective weight variant
    pair = pair_to_device({k[3:]: g[k] for k in g.files if k.startswith("in.")})
    with torch.no_grad():
        out = model.forward(**pair)
    for k in ("src_nodes", "tgt_nodes"):
        assert np.array_equal(out[k].cpu().numpy(), g["out." + k])
    for k in ("src_node_feats", "tgt_node_feats"):
        assert np.abs(out[k].cpu().numpy() - g["out." + k]).max() < 1e-4
    for k in ("src_point_feats", "tgt_point_feats"):
        err = np.abs(out[k].cpu().numpy()[::8] - g[f"out.{k}.every8"]).max()
        assert err < 1e-4, (k, err)
    assert np.array_equal(out["tgt_node_corr_indices"].cpu().numpy(), g["out.tgt_node_corr_indices"])
    assert np.array_equal(out["src_node_corr_indices"].cpu().numpy(), g["out.src_node_corr_indices"])
    ms, ref = out["matching_scores"].cpu().numpy()[::8], g["out.matching_scores.every8"]
    tm = np.concatenate([out["tgt_node_corr_knn_masks"].cpu().numpy()[::8], np.ones((ref.shape[0], 1), bool)], 1)
    sm = np.concatenate([out["src_node_corr_knn_masks"].cpu().numpy()[::8], np.ones((ref.shape[0], 1), bool)], 1)
    valid = tm[:, :, None] & sm[:, None, :]
    assert (np.abs(ms - ref) / np.maximum(1.0, np.abs(ref)))[valid].max() < 2e-4
    from corr_util import common_order_equal, compare_correspondences, to_numpy_corr
    got = to_numpy_corr(out)
    want = {k: g["out." + k] for k in ("tgt_corr_points", "src_corr_points", "corr_scores")}
    assert want["corr_scores"].shape[0] > 1000
    frac, err, _ = compare_correspondences(got, want)
    assert frac >= 0.995 and err < 1e-4, (frac, err)
    assert common_order_equal(got, want)
    np.testing.assert_allclose(out["gt_tgt_node_occ"].cpu().numpy(), g["out.gt_tgt_node_occ"], atol=1e-6)
