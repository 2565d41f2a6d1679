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


def test_patch_list_is_compacted_and_an_overfull_call_is_repeated_exactly():
    """Round 6: the 4DMatch tail (patch assembly, score contraction, optimal transport, fine matching) runs on the patches the
    adaptive matching SELECTED, laid out back to back over the pairs of the call (RIGA_v2.py:126-152 indexes with the selected
    node pairs only), in buffers sized by `patch_slots_per_pair` instead of the n4max^2 bound.  (1) Pairs of different sizes and
    selection counts in one call: every per-pair output bitwise equal to the pair run alone.  (2) A model whose buffers are far too
    small (4 slots per pair) warns, repeats the call with the exact count and returns the same bits.  (3) Both through the HIP-graph
    path.  (4) An explicit patch_slots at the exact total and at the n^2 bound give the same bits as well."""
    import warnings

    from roitr_amd.synthetic import make_pair
    from test_timed_shape_gpu import assert_bitwise
    sizes = (1500, 2600, 1024, 2048, 3100)
    pairs = [pair_to_device(make_pair(n, config=4, pair_index=40 + i, normals="field")) for i, n in enumerate(sizes)]
    model = build_model("4DMatch", weights="selective")
    with torch.no_grad():
        alone = [model.forward_batch([p])[0] for p in pairs]
        counts = [int(r["src_node_corr_indices"].shape[0]) for r in alone]
        assert len(set(counts)) > 1 and min(counts) >= 1, counts            # ragged selection counts: the prefix sums matter
        together = model.forward_batch(pairs)
        for i, (a, b) in enumerate(zip(together, alone)):
            assert_bitwise(a, b, f"pair {i} in a 5-pair call vs alone")
        total = sum(counts)
        h = model.launch_batch(pairs, patch_slots=total)                    # exactly full: no slot to spare, no repeat
        for i, (a, b) in enumerate(zip(model.finish_batch(h), alone)):
            assert_bitwise(a, b, f"pair {i}, patch_slots = the exact total")
        h = model.launch_batch(pairs, patch_slots=10 ** 9)                  # clamped to the B * n4max^2 bound
        for i, (a, b) in enumerate(zip(model.finish_batch(h), alone)):
            assert_bitwise(a, b, f"pair {i}, patch_slots = the bound")
        small = build_model("4DMatch", weights="selective")
        small.patch_slots_per_pair = 4
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = small.forward_batch(pairs)
        assert any("repeating the call" in str(x.message) for x in w), [str(x.message) for x in w]
        for i, (a, b) in enumerate(zip(got, alone)):
            assert_bitwise(a, b, f"pair {i} after the repeated call")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            for _ in range(3):                                              # warm-up, capture, replay: each overfull, each repeated on the plain path
                got = small.forward_batch(pairs[:2], graph=True)
                for i, (a, b) in enumerate(zip(got, alone[:2])):
                    assert_bitwise(a, b, f"pair {i} through the graph path with overfull buffers")
        for _ in range(3):
            got = model.forward_batch(pairs[:2], graph=True)
            for i, (a, b) in enumerate(zip(got, alone[:2])):
                assert_bitwise(a, b, f"pair {i} through the graph path")
