"""GPU parity: HIP pointops (through the C ABI) vs the CPU oracle and the golden fixtures.

Bar: bit-exact int32 indices and bit-exact fp32 squared distances for FPS / kNN (both follow the
same documented fmaf form); PPF within 2e-6 absolute (atan2f/sqrtf device libm vs torch CPU).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pointops_cpu as O  # noqa: E402  (checker only)


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def cloud(rng, n, kind="uniform"):
    if kind == "uniform":
        return (rng.random((n, 3)) * 2).astype(np.float32)
    if kind == "lattice":  # heavy exact ties
        g = rng.integers(0, 6, (n, 3)).astype(np.float32) * 0.25
        return g
    if kind == "dup":  # duplicated points
        base = (rng.random((max(n // 3, 1), 3)) * 2).astype(np.float32)
        return base[rng.integers(0, len(base), n)]
    if kind == "plane":
        p = (rng.random((n, 3)) * 2).astype(np.float32)
        p[:, 2] = 0.5
        return p
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["uniform", "lattice", "dup", "plane"])
@pytest.mark.parametrize("sizes", [[5000], [1250], [312], [78], [16], [1024, 700, 2049], [8000], [3, 1, 130], [30000, 17000], [31000]])
def test_fps_bit_exact(kind, sizes):
    from roitr_amd import pointops as P
    rng = np.random.default_rng(1000 * len(kind) + sum(sizes))   # not hash(): str hashes differ from process to process
    xyz = np.concatenate([cloud(rng, n, kind) for n in sizes])
    off = np.cumsum(sizes).astype(np.int32)
    noff = np.cumsum([max(n // 4, 1) for n in sizes]).astype(np.int32)
    ref = O.furthestsampling(xyz, off, noff)
    got = P.furthestsampling(dev(xyz), dev(off), dev(noff)).cpu().numpy()
    assert got.dtype == np.int32 and np.array_equal(got, ref)


@pytest.mark.parametrize("kind", ["uniform", "lattice"])
@pytest.mark.parametrize("n_clouds,n", [(40, 1250), (24, 5000), (33, 313)])
def test_fps_bit_exact_many_clouds(kind, n_clouds, n):
    """More than 16 clouds per call take the 4-wave workgroups of the batched forward, 16 or fewer the 8-wave ones of the one-pair
    mode (pointops_fps.hip launcher): the indices are the same either way, and the same as the oracle's."""
    from roitr_amd import pointops as P
    rng = np.random.default_rng(1000 * len(kind) + 31 * n_clouds + n)
    sizes = [n - (i % 3) for i in range(n_clouds)]
    xyz = np.concatenate([cloud(rng, m, kind) for m in sizes])
    off = np.cumsum(sizes).astype(np.int32)
    noff = np.cumsum([max(m // 4, 1) for m in sizes]).astype(np.int32)
    ref = O.furthestsampling(xyz, off, noff)
    got = P.furthestsampling(dev(xyz), dev(off), dev(noff)).cpu().numpy()
    assert np.array_equal(got, ref)
    # the first 8 clouds alone (8-wave workgroups) reproduce their part of the batched call
    few = P.furthestsampling(dev(xyz[:off[7]]), dev(off[:8]), dev(noff[:8])).cpu().numpy()
    assert np.array_equal(few, ref[:noff[7]])


@pytest.mark.parametrize("kind", ["uniform", "lattice", "dup"])
@pytest.mark.parametrize("sizes", [[5000, 4999, 1250], [1024] * 20, [313, 5000]])
def test_fps_hierarchy_chain_with_prefix_shortcut(kind, sizes):
    """roitr_furthestsampling_ex over three levels the way the engine chains it (each level samples the previous level's picks
    in pick order): clouds whose tracked arg-maxima were unique are answered with the prefix, the others (lattice / duplicate
    clouds: shared maxima everywhere) run the chain -- either way every level equals the oracle's plain FPS of that level."""
    import ctypes
    from roitr_amd import _lib as L
    rng = np.random.default_rng(1000 * len(kind) + sum(sizes))   # not hash(): str hashes differ from process to process
    xyz = np.concatenate([cloud(rng, n, kind) for n in sizes])
    b = len(sizes)
    lib = L.lib()
    cur, cur_sizes = xyz, list(sizes)
    prev_tie = None
    shortcut_taken = []
    for lvl in range(3):
        nxt_sizes = [max(n // 4, 1) for n in cur_sizes]
        off = np.cumsum(cur_sizes).astype(np.int32)
        noff = np.cumsum(nxt_sizes).astype(np.int32)
        ref = O.furthestsampling(cur, off, noff)
        d_xyz, d_off, d_noff = dev(cur), dev(off), dev(noff)
        tmp = torch.full((cur.shape[0],), 1e10, dtype=torch.float32, device="cuda")
        idx = torch.empty((int(noff[-1]),), dtype=torch.int32, device="cuda")
        tie = torch.full((b,), -7, dtype=torch.int32, device="cuda")
        L.check(lib.roitr_furthestsampling_ex(b, int(max(cur_sizes)), L.ptr(d_xyz), L.ptr(d_off), L.ptr(d_noff), L.ptr(tmp), L.ptr(idx),
                                              L.ptr(prev_tie), L.ptr(tie), 4, L.stream_ptr()), "fps_ex")
        got = idx.cpu().numpy()
        assert np.array_equal(got, ref), (kind, lvl)
        if prev_tie is not None:
            pt = prev_tie.cpu().numpy()
            shortcut_taken.append([bool(pt[c] >= nxt_sizes[c]) for c in range(b)])
        prev_tie = tie
        cur = np.ascontiguousarray(cur[got.astype(np.int64)])
        cur_sizes = nxt_sizes
    if kind == "uniform":
        # shared maxima are rare on a generic cloud (an exact fp32 distance tie among the tracked picks): most clouds take the prefix
        taken = [t for r in shortcut_taken for t in r]
        assert sum(taken) * 2 >= len(taken)
    if kind == "lattice":
        assert not all(all(r) for r in shortcut_taken)      # the fallback really ran somewhere


def test_fps_golden(golden_pair):
    from roitr_amd import pointops as P
    g = golden_pair
    for side, base in (("src", 0), ("tgt", 3)):
        p = dev(g[f"in.{side}_points"] if side == "tgt" else g["in.raw_src_pcd"])
        n = p.shape[0]
        for lvl in range(3):
            o = torch.tensor([n], dtype=torch.int32).cuda()
            no = torch.tensor([n // 4], dtype=torch.int32).cuda()
            idx = P.furthestsampling(p, o, no)
            assert np.array_equal(idx.cpu().numpy(), g[f"fps.{base + lvl}"])
            p = p[idx.long()].contiguous()
            n = n // 4


@pytest.mark.parametrize("use_grid", [False, True])
@pytest.mark.parametrize("kind", ["uniform", "lattice", "dup", "plane"])
@pytest.mark.parametrize("case", [
    # (ref sizes, query = self?, nsample)
    ([5000], True, 9), ([5000], True, 17), ([1250], True, 17), ([312], True, 17), ([78], True, 17),
    ([16], True, 17), ([5000], False, 17), ([5000], False, 3), ([5000], False, 1), ([900, 1500, 40], True, 9),
    ([900, 1500, 40], False, 17), ([3000], True, 65), ([2000], True, 100), ([5], True, 3),
    # round 6: more shapes through the cell kernel (self queries on a grid below 8192 queries): other k, two clouds, the largest such cloud
    ([8000], True, 33), ([700, 3100], True, 49), ([8100], True, 65),
])
def test_knn_bit_exact(case, kind, use_grid):
    from roitr_amd import pointops as P
    sizes, self_q, ns = case
    rng = np.random.default_rng(1000 * len(kind) + sum(sizes) + 7 * int(self_q) + 13 * ns)
    xyz = np.concatenate([cloud(rng, n, kind) for n in sizes])
    off = np.cumsum(sizes).astype(np.int32)
    if self_q:
        q, qoff = xyz, off
    else:
        qs = [max(n // 4, 1) for n in sizes]
        # queries partly outside the reference bounding box
        q = np.concatenate([cloud(rng, m, "uniform") * 1.3 - 0.3 for m in qs]).astype(np.float32)
        qoff = np.cumsum(qs).astype(np.int32)
    ridx, rd2 = O.knnquery_raw(ns, xyz, q, off, qoff, threads=8)
    idx, d2 = P.knnquery_raw(ns, dev(xyz), dev(q), dev(off), dev(qoff), use_grid=use_grid)
    idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    assert np.array_equal(d2, rd2), f"dist2 mismatch rows={np.unique(np.nonzero(d2 != rd2)[0])[:10]}"
    assert np.array_equal(idx, ridx), f"idx mismatch rows={np.unique(np.nonzero(idx != ridx)[0])[:10]}"


@pytest.mark.parametrize("kind", ["uniform", "lattice", "dup", "plane", "surface"])
@pytest.mark.parametrize("case", [
    # (clouds, points per cloud, query = self?, nsample): every case has >= 8192 queries on a grid, i.e. the LANE-per-query kernels
    # of the batched forward (prefilter for nsample + 1 <= 10, its retry list -> ring kernel, ring kernel alone above, -> replay)
    (4, 5000, True, 9), (4, 5000, True, 17), (8, 1250, True, 17), (3, 5000, False, 17), (9, 1250, False, 17),
    (4, 5000, False, 3), (4, 5000, False, 1), (2, 9000, True, 5), (1, 70000, True, 9),
])
def test_knn_lane_kernels_bit_exact(case, kind):
    """The kernels the engine's 512-pair calls run (one lane per query) against the C oracle: idx and dist2 bit-equal, clouds of
    slightly different sizes, queries of non-self calls partly outside the reference cloud's box; one 70000-point cloud (more
    points than a 16-bit position holds, the largest grid)."""
    from roitr_amd import pointops as P
    nc, n, self_q, ns = case
    if kind != "uniform" and n > 9000:
        pytest.skip("one large case is enough")
    rng = np.random.default_rng(77 * len(kind) + 1000 * nc + n + 7 * int(self_q) + 13 * ns)
    sizes = [n - 3 * (i % 4) for i in range(nc)]

    def make(m):
        if kind == "surface":   # a noisy two-plane corner: cells mostly empty, the occupied ones dense
            p = (rng.random((m, 3)) * 2).astype(np.float32)
            half = m // 2
            p[:half, 2] = 0.3 + 0.004 * rng.standard_normal(half).astype(np.float32)
            p[half:, 0] = 0.1 + 0.004 * rng.standard_normal(m - half).astype(np.float32)
            return p
        return cloud(rng, m, kind)

    xyz = np.concatenate([make(m) for m in sizes])
    off = np.cumsum(sizes).astype(np.int32)
    if self_q:
        q, qoff = xyz, off
    else:
        # FPS-like subsets would be self points: use fresh points, 1/3 of them outside the reference box
        qs = [max(8192 // nc + 17, m // 2) for m in sizes]
        q = np.concatenate([make(m) * (1.3 if i % 3 == 0 else 1.0) - (0.3 if i % 3 == 0 else 0.0) for i, m in enumerate(qs)]).astype(np.float32)
        qoff = np.cumsum(qs).astype(np.int32)
    assert q.shape[0] >= 8192
    ridx, rd2 = O.knnquery_raw(ns, xyz, q, off, qoff, threads=8)
    idx, d2 = P.knnquery_raw(ns, dev(xyz), dev(q), dev(off), dev(qoff), use_grid=True)
    idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    assert np.array_equal(d2, rd2), f"dist2 mismatch rows={np.unique(np.nonzero(d2 != rd2)[0])[:10]}"
    assert np.array_equal(idx, ridx), f"idx mismatch rows={np.unique(np.nonzero(idx != ridx)[0])[:10]}"


def test_knn_within_matches_oracle_below_the_cap():
    """roitr_knn_within (the ground-truth occlusion test, lib/utils.py:509-521): nearest squared distance exact wherever it is below the
    cap, some value >= the cap elsewhere -- lane-per-query path (>= 8192 queries) and wave path."""
    from roitr_amd import _lib as L
    rng = np.random.default_rng(9)
    for nc, n in ((6, 5000), (1, 3000)):
        sizes = [n - i for i in range(nc)]
        xyz = np.concatenate([cloud(rng, m) for m in sizes])
        q = np.concatenate([cloud(rng, m) * 1.1 - 0.05 for m in sizes]).astype(np.float32)
        off = np.cumsum(sizes).astype(np.int32)
        _, rd2 = O.knnquery_raw(1, xyz, q, off, off, threads=8)
        for cap in (0.0375, 0.08, 0.5):
            cap2 = cap * cap * 1.01
            lib = L.lib()
            b, nn, m = len(sizes), xyz.shape[0], q.shape[0]
            txyz, tq, toff = dev(xyz), dev(q), dev(off)
            ws = torch.empty(lib.roitr_knn_workspace_bytes(b, nn, m), dtype=torch.uint8, device="cuda")
            d2 = torch.full((m,), -1.0, dtype=torch.float32, device="cuda")
            L.check(lib.roitr_knn_build_grid(b, nn, m, L.ptr(txyz), L.ptr(toff), L.ptr(ws), L.stream_ptr()), "grid")
            import ctypes
            L.check(lib.roitr_knn_within(b, nn, m, L.ptr(txyz), L.ptr(tq), L.ptr(toff), L.ptr(toff), ctypes.c_float(cap2), L.ptr(d2), 1, m,
                                         L.ptr(ws), L.stream_ptr()), "within")
            got = d2.cpu().numpy()
            below = rd2[:, 0] < cap2
            assert below.sum() > 10
            assert np.array_equal(got[below], rd2[below, 0])
            assert (got[~below] >= cap2).all()


def test_ppf_against_float64():
    """The shared PPF arithmetic (common.h roitr_ppf4: polynomial atan2, v_rcp / v_sqrt) against a float64 evaluation of
    lib/utils.py:358-389 on random, near-parallel, anti-parallel and zero vectors: 5e-7 on angles / pi, 1e-6 relative on |d|."""
    from roitr_amd import ops
    rng = np.random.default_rng(21)
    m, k = 4096, 16
    pts = rng.standard_normal((m, 3)).astype(np.float32)
    nrm = rng.standard_normal((m, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    patches = (pts[:, None, :] + 0.2 * rng.standard_normal((m, k, 3))).astype(np.float32)
    pn = rng.standard_normal((m, k, 3)).astype(np.float32)
    pn /= np.linalg.norm(pn, axis=2, keepdims=True)
    pn[:, 0] = nrm                                                        # parallel normals
    pn[:, 1] = -nrm                                                       # anti-parallel
    patches[:, 2] = pts + nrm * np.float32(0.1)                           # d parallel to n1
    patches[:, 3] = pts                                                   # d = 0
    pn[:, 4] = nrm + np.float32(1e-4) * rng.standard_normal((m, 3)).astype(np.float32)   # nearly parallel
    pn[:, 5] = 0.0                                                        # zero normal
    grp = torch.arange(m * k, dtype=torch.int32).view(m, k).cuda()
    out = ops.calc_ppf(dev(pts), dev(nrm), dev(patches.reshape(-1, 3)), dev(pn.reshape(-1, 3)), grp).cpu().numpy()
    P64, N64, Q64, M64 = pts.astype(np.float64), nrm.astype(np.float64), patches.astype(np.float64), pn.astype(np.float64)
    d = Q64 - P64[:, None, :]

    def ang(a, b):
        return np.arctan2(np.linalg.norm(np.cross(a, b), axis=-1), (a * b).sum(-1)) / np.pi

    n1 = np.broadcast_to(N64[:, None, :], d.shape)
    ref = np.stack([np.linalg.norm(d, axis=-1), ang(n1, d), ang(M64, d), ang(n1, M64)], -1)
    assert np.abs(out[..., 0] - ref[..., 0]).max() < 1e-6
    # near-degenerate pairs amplify the fp32 rounding of the cross / dot products themselves (any fp32 implementation does):
    # compare where the float64 operands are not within 1e-3 rad of (anti-)parallel, and those loosely
    err = np.abs(out[..., 1:] - ref[..., 1:])
    generic = (ref[..., 1:] > 1e-3) & (ref[..., 1:] < 1 - 1e-3)
    assert err[generic].max() < 5e-7, err[generic].max()
    assert err.max() < 2e-4


def test_knn_golden(golden_pair):
    from roitr_amd import pointops as P
    g = golden_pair
    # first calls of the forward: enc1 TransitionDown self-kNN(9) on the raw source cloud
    p = dev(g["in.raw_src_pcd"])
    o = torch.tensor([p.shape[0]], dtype=torch.int32).cuda()
    idx, dist = P.knnquery(9, p, p, o, o)
    assert np.array_equal(idx.cpu().numpy(), g["knn.0.idx"])
    # torch-CPU sqrt (Sleef) is not always correctly rounded: 1-ulp slack on the euclidean distances only
    np.testing.assert_allclose(dist.cpu().numpy(), g["knn.0.dist"], rtol=2e-7, atol=0)


def test_knn_ppf_fused_matches_golden(golden_pair):
    from roitr_amd import pointops as P
    g = golden_pair
    p, n = dev(g["in.raw_src_pcd"]), dev(g["in.src_normals"])
    o = torch.tensor([p.shape[0]], dtype=torch.int32).cuda()
    grp, ppf = P.knn_ppf(8, p, p, n, n, o, o)
    assert np.array_equal(grp.cpu().numpy(), g["knn.0.idx"][:, 1:])
    np.testing.assert_allclose(ppf.cpu().numpy(), g["ppf.0"], rtol=0, atol=2e-6)


def test_ppf_stage_golden(golden_stages):
    """calc_ppf_gpu known answers incl. the atan2(0,0) and parallel-normal corner cases."""
    from roitr_amd import pointops as P
    s = golden_stages
    pts, nrm, patches, pn = s["ppf.pts"], s["ppf.nrm"], s["ppf.patches"], s["ppf.pnrm"]
    m, k, _ = patches.shape
    # lay the patches out as a reference cloud and query with explicit group indices via kNN-free path:
    # reference cloud = [patches reshaped], each centre's neighbours are its own k rows
    from roitr_amd import ops
    grp = torch.arange(m * k, dtype=torch.int32).view(m, k).cuda()
    out = ops.calc_ppf(dev(pts), dev(nrm), dev(patches.reshape(-1, 3)), dev(pn.reshape(-1, 3)), grp)
    np.testing.assert_allclose(out.cpu().numpy(), s["ppf.out"], rtol=0, atol=2e-6)


def test_interpolation_matches_oracle():
    from roitr_amd import pointops as P
    rng = np.random.default_rng(5)
    xyz = cloud(rng, 312)
    new = cloud(rng, 1250)
    feat = rng.normal(0, 1, (312, 64)).astype(np.float32)
    o, no = np.array([312], np.int32), np.array([1250], np.int32)
    ref = O.interpolation(xyz, new, feat, o, no)
    got = P.interpolation(dev(xyz), dev(new), dev(feat), dev(o), dev(no)).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)


def test_cold_ops_match_oracle():
    from roitr_amd import pointops as P
    rng = np.random.default_rng(11)
    n, ns, c, wc = 200, 8, 32, 8
    inp = rng.normal(0, 1, (n, c)).astype(np.float32)
    inp2 = rng.normal(0, 1, (n, c)).astype(np.float32)
    pos = rng.normal(0, 1, (n, ns, c)).astype(np.float32)
    w = rng.normal(0, 1, (n, ns, wc)).astype(np.float32)
    idx = rng.integers(0, n, (n, ns)).astype(np.int32)
    g3 = rng.normal(0, 1, (n, ns, c)).astype(np.float32)
    g2 = rng.normal(0, 1, (n, c)).astype(np.float32)
    tin, tin2, tpos, tw, tidx = dev(inp), dev(inp2), dev(pos), dev(w), dev(idx)

    x = tin.clone().requires_grad_(True)
    out = P.grouping(x, tidx)
    assert np.array_equal(out.detach().cpu().numpy(), O.grouping_forward(inp, idx))
    out.backward(dev(g3))
    np.testing.assert_allclose(x.grad.cpu().numpy(), O.grouping_backward(g3, idx, n), rtol=1e-5, atol=1e-5)

    a, b = tin.clone().requires_grad_(True), tin2.clone().requires_grad_(True)
    out = P.subtraction(a, b, tidx)
    assert np.array_equal(out.detach().cpu().numpy(), O.subtraction_forward(inp, inp2, idx))
    out.backward(dev(g3))
    r1, r2 = O.subtraction_backward(idx, g3)
    np.testing.assert_allclose(a.grad.cpu().numpy(), r1, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(b.grad.cpu().numpy(), r2, rtol=1e-5, atol=1e-5)

    a, p_, w_ = tin.clone().requires_grad_(True), tpos.clone().requires_grad_(True), tw.clone().requires_grad_(True)
    out = P.aggregation(a, p_, w_, tidx)
    np.testing.assert_allclose(out.detach().cpu().numpy(), O.aggregation_forward(inp, pos, w, idx), rtol=1e-5, atol=1e-5)
    out.backward(dev(g2))
    ri, rp, rw = O.aggregation_backward(inp, pos, w, idx, g2)
    np.testing.assert_allclose(a.grad.cpu().numpy(), ri, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(p_.grad.cpu().numpy(), rp, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(w_.grad.cpu().numpy(), rw, rtol=1e-4, atol=1e-4)

    # interpolation2 (native forward/backward)
    xyz, new = cloud(rng, 100), cloud(rng, 300)
    feat = rng.normal(0, 1, (100, 16)).astype(np.float32)
    o, no = np.array([100], np.int32), np.array([300], np.int32)
    f = dev(feat).requires_grad_(True)
    out = P.interpolation2(dev(xyz), dev(new), f, dev(o), dev(no), 3)
    np.testing.assert_allclose(out.detach().cpu().numpy(), O.interpolation(xyz, new, feat, o, no), rtol=1e-5, atol=1e-6)
    go = rng.normal(0, 1, (300, 16)).astype(np.float32)
    out.backward(dev(go))
    kidx, kd = O.knnquery(3, xyz, new, o, no)
    rec = 1.0 / (kd + 1e-8)
    wgt = (rec / rec.sum(1, keepdims=True)).astype(np.float32)
    np.testing.assert_allclose(f.grad.cpu().numpy(), O.interpolation_backward(go, kidx, wgt, 100), rtol=1e-4, atol=1e-5)


def test_legacy_launchers_exact_names():
    """The reference's own extern "C" symbols (void, default stream) work as drop-ins."""
    import ctypes
    from roitr_amd import _lib as L
    rng = np.random.default_rng(3)
    xyz = cloud(rng, 700)
    off = np.array([300, 700], np.int32)
    noff = np.array([75, 175], np.int32)
    lib = L.lib()
    txyz, toff, tnoff = dev(xyz), dev(off), dev(noff)
    idx = torch.zeros(175, dtype=torch.int32).cuda()
    tmp = torch.full((700,), 1e10).cuda()
    torch.cuda.synchronize()
    lib.furthestsampling_cuda_launcher(2, 400, L.ptr(txyz), L.ptr(toff), L.ptr(tnoff), L.ptr(tmp), L.ptr(idx))
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy(), O.furthestsampling(xyz, off, noff))
    new = txyz[idx.long()].contiguous()
    kidx = torch.zeros((175, 17), dtype=torch.int32).cuda()
    kd2 = torch.zeros((175, 17), dtype=torch.float32).cuda()
    torch.cuda.synchronize()
    lib.knnquery_cuda_launcher(175, 17, L.ptr(txyz), L.ptr(new), L.ptr(toff), L.ptr(tnoff), L.ptr(kidx), L.ptr(kd2))
    torch.cuda.synchronize()
    ridx, rd2 = O.knnquery_raw(17, xyz, new.cpu().numpy(), off, noff)
    assert np.array_equal(kidx.cpu().numpy(), ridx) and np.array_equal(kd2.cpu().numpy(), rd2)
