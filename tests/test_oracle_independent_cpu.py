"""CPU: independent cross-checks of the UNPINNED native oracle (oracle/pointops_ref.c; the CUDA original cannot run here).

1. An independently written float64 brute force (numpy only, no code shared with pointops_ref.c) must agree with the oracle's
   FPS / kNN indices on tie-free clouds: every FPS pick is the float64 arg-max of the running minimum distance (to within fp32
   rounding of the two leading candidates), every kNN row equals the float64 stable sort (mismatches only where two float64
   distances agree to fp32 rounding).
2. The oracle's documented arithmetic policy is nvcc's default contraction fmaf(dz,dz,fmaf(dy,dy,dx*dx))
   (sampling_cuda_kernel.cu:49-59, knnquery_cuda_kernel.cu:92-102); the original binary's contraction cannot be inspected.
   Recomputing every pinned cloud (golden pairs, all four hierarchy levels, and bench-workload pairs) with the UNCONTRACTED
   fp32 form (dx*dx + dy*dy) + dz*dz shows the policy is unobservable there: the oracle's FPS picks are the unique
   arg-maxima and its kNN rows the strictly ordered minima under the uncontracted distances as well.
"""
import os

import numpy as np
import pytest

from oracle import pointops_cpu as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
F32 = np.float32


# ---------------------------------------------------------------- 1. float64 brute force
def fps_replay_f64(xyz, idx):
    """For the pick sequence `idx`: (#picks that are the float64 arg-max, worst relative shortfall of a pick)."""
    p = xyz.astype(np.float64)
    best = np.full(len(p), np.inf)
    exact, worst = 0, 0.0
    assert idx[0] == 0                       # first index = segment start (sampling_cuda_kernel.cu:30-31)
    for j in range(1, len(idx)):
        d = ((p - p[idx[j - 1]]) ** 2).sum(1)
        best = np.minimum(best, d)
        top = best.max()
        exact += int(best.argmax() == idx[j])
        worst = max(worst, (top - best[idx[j]]) / top)
    return exact, worst


def knn_f64(xyz, q, k):
    p, qq = xyz.astype(np.float64), q.astype(np.float64)
    idx = np.empty((len(qq), k), np.int64)
    d2 = np.empty((len(qq), k))
    for lo in range(0, len(qq), 512):
        d = ((qq[lo:lo + 512, None, :] - p[None, :, :]) ** 2).sum(2)
        o = np.argsort(d, axis=1, kind="stable")[:, :k]
        idx[lo:lo + 512] = o
        d2[lo:lo + 512] = np.take_along_axis(d, o, 1)
    return idx, d2


@pytest.mark.parametrize("n,seed", [(5000, 1), (1250, 2), (3001, 3)])
def test_fps_matches_independent_float64(n, seed):
    xyz = (np.random.default_rng(seed).random((n, 3)) * 2).astype(F32)
    m = n // 4
    idx = O.furthestsampling(xyz, np.array([n], np.int32), np.array([m], np.int32))
    assert len(set(idx.tolist())) == m
    exact, worst = fps_replay_f64(xyz, idx)
    assert worst < 1e-6                      # a pick that is not THE float64 arg-max loses to it by fp32 rounding at most
    assert exact >= (m - 1) - 2, (exact, m)


@pytest.mark.parametrize("n,m,k,seed", [(5000, 5000, 17, 4), (5000, 1250, 9, 5), (3000, 3000, 65, 6), (1250, 5000, 3, 7)])
def test_knn_matches_independent_float64(n, m, k, seed):
    rng = np.random.default_rng(seed)
    xyz = (rng.random((n, 3)) * 2).astype(F32)
    q = xyz if m == n else (rng.random((m, 3)) * 2).astype(F32)
    o, qo = np.array([n], np.int32), np.array([m], np.int32)
    idx, d2 = O.knnquery_raw(k, xyz, q, o, qo, threads=8)
    ridx, rd2 = knn_f64(xyz, q, k)
    np.testing.assert_allclose(d2, rd2, rtol=3e-7, atol=1e-12)      # fp32 evaluation of the same distances
    bad = np.nonzero(idx != ridx)
    for r, c in zip(*bad):                    # only where float64 sees a near-tie that fp32 rounding can reorder
        assert abs(rd2[r, c] - rd2[r, np.nonzero(ridx[r] == idx[r, c])[0][0]]) <= 4e-7 * rd2[r, c], (r, c)
    assert len(bad[0]) <= 1e-4 * idx.size


# ---------------------------------------------------------------- 2. uncontracted fp32 form
def d2_uncontracted(p, c):
    d = p - c                                  # fp32
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]   # every op rounds to fp32: no FMA


def fps_unobservable(xyz, idx):
    best = np.full(len(xyz), F32(1e10), F32)   # tmp = 1e10 (functions/pointops.py:22)
    for j in range(1, len(idx)):
        best = np.minimum(best, d2_uncontracted(xyz, xyz[idx[j - 1]]))
        top = best.max()
        if best[idx[j]] != top or (best == top).sum() != 1:
            return False
    return True


def knn_unobservable(xyz, q, idx):
    """rows of `idx` that are NOT the strictly ordered k minima under the uncontracted distances"""
    bad = 0
    k = min(idx.shape[1], len(xyz))            # a cloud smaller than k leaves (1e10, first row) fill entries behind
    idx = idx[:, :k]
    for lo in range(0, len(q), 1024):
        qs = q[lo:lo + 1024]
        d = qs[:, None, :] - xyz[None, :, :]
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        sel = np.take_along_axis(d2, idx[lo:lo + 1024].astype(np.int64), 1)
        ordered = (np.diff(sel, axis=1) > 0).all(1)
        rest = d2.copy()
        np.put_along_axis(rest, idx[lo:lo + 1024].astype(np.int64), np.inf, 1)
        bad += int((~(ordered & (rest.min(1) > sel[:, k - 1]))).sum())
    return bad


def pinned_clouds():
    from roitr_amd.synthetic import make_pair
    out = []
    for name in ("pair_n1024.npz", "pair_4dmatch_n1024.npz"):
        g = np.load(os.path.join(GOLD, name))
        out += [(name + ":src", g["in.raw_src_pcd"]), (name + ":tgt", g["in.tgt_points"])]
    for i in (0, 1):
        p = make_pair(5000, config=2, pair_index=i)
        out += [(f"bench{i}:src", p["raw_src_pcd"]), (f"bench{i}:tgt", p["tgt_points"])]
    return out


def test_fma_policy_is_unobservable_on_the_pinned_clouds():
    nsample = [8, 16, 16, 16]
    for name, p in pinned_clouds():
        for lvl in range(4):
            n = p.shape[0]
            o = np.array([n], np.int32)
            if lvl > 0:                        # TransitionDown: FPS n -> n//4, then kNN of the sampled points in the finer cloud
                m = n // 4
                idx = O.furthestsampling(p, o, np.array([m], np.int32))
                assert fps_unobservable(p, idx), (name, lvl)
                q = p[idx]
                kid, _ = O.knnquery_raw(nsample[lvl] + 1, p, q, o, np.array([m], np.int32), threads=8)
                assert knn_unobservable(p, q, kid) == 0, (name, lvl, "td")
                p = np.ascontiguousarray(q)
                o = np.array([m], np.int32)
            kid, _ = O.knnquery_raw(nsample[lvl] + 1, p, p, o, o, threads=8)
            assert knn_unobservable(p, p, kid) == 0, (name, lvl, "self")
            if lvl > 0:                        # decoder 3-NN of the finer level in this one is a subset of the same arithmetic
                pass
