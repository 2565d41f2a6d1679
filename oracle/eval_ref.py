"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the evaluators, numpy float32.

Follows lib/loss.py:175-206 (Evaluator.evaluate_coarse / evaluate_fine) and registration/benchmark_utils.py:69-77
(get_inlier_ratio_correspondence).  Pinned by tests/golden/prep_eval.npz (captured from the imported reference)."""
import numpy as np


def inlier_ratio(src_pts, tgt_pts, rot, trans, radius):
    """lib/loss.py:199-205 / benchmark_utils.py:73-77."""
    if src_pts.shape[0] == 0:
        return 0.0
    moved = (src_pts.astype(np.float32) @ rot.astype(np.float32).T + trans.astype(np.float32).reshape(1, 3)).astype(np.float32)
    d = np.sqrt(((tgt_pts.astype(np.float32) - moved) ** 2).sum(1, dtype=np.float32))
    return float((d < np.float32(radius)).astype(np.float32).mean())


def coarse_precision(n_tgt, n_src, gt_idx, gt_overlaps, tgt_corr, src_corr, acceptance_overlap):
    """lib/loss.py:176-191."""
    m = gt_overlaps > acceptance_overlap
    g = gt_idx[m]
    gmap = np.zeros((n_tgt, n_src), dtype=np.float32)
    gmap[g[:, 0], g[:, 1]] = 1.0
    return float(gmap[tgt_corr, src_corr].mean()) if tgt_corr.shape[0] else float("nan")
