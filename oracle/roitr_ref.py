"""CPU restatement (numpy, fp32) of the RoITr test-mode forward -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  It is the
checker for the HIP path, never the thing measured or shipped.

Each function cites the reference lines it restates (paths relative to /root/reference).  Unlike the HIP
engine it is NOT algebraically folded: it forms proj_p(embedding(ppf)), proj_p(E), the (M,K,C) gathers etc.
exactly as the reference does, so it also checks the folds.

Pinning: validated in tests/test_oracle_cpu.py against tensors captured from the imported reference
(tests/golden/pair_n1024.npz, stages.npz).  FPS / kNN come from oracle/pointops_ref.c, whose parity with the
CUDA-only original is unpinned (see that file's header).
"""
import os
import time

import numpy as np

from . import pointops_cpu as P

f32 = np.float32


# ------------------------------------------------------------------------------------------ primitives
def linear(x, W, b=None):
    y = x.astype(f32) @ W.T.astype(f32)
    if b is not None:
        y = y + b
    return y.astype(f32)


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True, dtype=f32)
    var = ((x - mu) ** 2).mean(-1, keepdims=True, dtype=f32)
    return ((x - mu) / np.sqrt(var + f32(eps)) * w + b).astype(f32)


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=axis, keepdims=True)).astype(f32)


def relu(x):
    return np.maximum(x, 0).astype(f32)


def atan2_pos(x, y):
    return np.arctan2(x, y).astype(f32)


def calc_ppf(points, point_normals, patches, patch_normals):
    """lib/utils.py:358-389"""
    pts = points[:, None, :]
    pn = np.broadcast_to(point_normals[:, None, :], patches.shape)
    vec_d = (patches - pts).astype(f32)
    d = np.sqrt((vec_d ** 2).sum(-1, keepdims=True, dtype=f32))

    def ang(a, b):
        y = (a * b).sum(-1, keepdims=True, dtype=f32)
        x = np.sqrt((np.cross(a, b).astype(f32) ** 2).sum(-1, keepdims=True, dtype=f32))
        return (atan2_pos(x, y) / f32(np.pi)).astype(f32)
    return np.concatenate([d, ang(pn, vec_d), ang(patch_normals, vec_d), ang(pn, patch_normals)], -1).astype(f32)


def square_distance(src, tgt, normalized=False):
    """lib/utils.py:139-156 for (N,C),(M,C)"""
    if normalized:
        dist = f32(2.0) - f32(2.0) * (src @ tgt.T)
    else:
        dist = f32(-2.0) * (src @ tgt.T)
        dist = dist + (src ** 2).sum(-1, dtype=f32)[:, None]
        dist = dist + (tgt ** 2).sum(-1, dtype=f32)[None, :]
    return np.maximum(dist, f32(1e-12)).astype(f32)


class Weights:
    def __init__(self, sd):
        self.sd = sd

    def lin(self, x, prefix):
        return linear(x, self.sd[prefix + ".weight"], self.sd[prefix + ".bias"])

    def ln(self, x, prefix):
        return layer_norm(x, self.sd[prefix + ".weight"], self.sd[prefix + ".bias"])


# ------------------------------------------------------------------------------------------ local PPF transformer
def local_ppf_transformer(W, pre, feats, node_idx, group_idx, ppf, heads=4):
    """LocalPPFTransformer.forward, ppftransformer.py:227-253 + attention.py:152-200, 298-320"""
    pos = W.lin(ppf, pre + ".embedding.proj")                     # positional_encoding.py:78-79
    f = W.lin(feats, pre + ".in_proj")
    at = pre + ".transformer.attention"
    q, k, v = W.lin(f, at + ".proj_q"), W.lin(f, at + ".proj_k"), W.lin(f, at + ".proj_v")
    p, vp = W.lin(pos, at + ".proj_p"), W.lin(pos, at + ".proj_vp")
    M, K = group_idx.shape
    H = q.shape[1]
    c = H // heads
    qn = q[node_idx].reshape(M, heads, 1, c)
    kg = k[group_idx].reshape(M, K, heads, c).transpose(0, 2, 1, 3)
    vg = v[group_idx].reshape(M, K, heads, c).transpose(0, 2, 1, 3)
    pg = p.reshape(M, K, heads, c).transpose(0, 2, 1, 3)
    vpg = vp.reshape(M, K, heads, c).transpose(0, 2, 1, 3)
    s_p = np.einsum("bhnc,bhmc->bhnm", qn, pg)
    s_e = np.einsum("bhnc,bhmc->bhnm", qn, kg)
    a = softmax((s_e + s_p) / f32(c ** 0.5), -1)
    hid = np.matmul(a, vg + vpg).transpose(0, 2, 1, 3).reshape(M, H).astype(f32)
    hid = W.lin(hid, pre + ".transformer.linear")
    out = W.ln(hid + f[node_idx], pre + ".transformer.norm")
    return W.lin(out, pre + ".out_proj")


def block(W, pre, x, group_idx, ppf):
    """RIPointTransformerBlock.forward, model/model.py:131-142"""
    n = x.shape[0]
    y = local_ppf_transformer(W, pre + ".transformer.transformer", x, np.arange(n), group_idx, ppf)
    return relu(W.ln(y, pre + ".bn2") + x)


# ------------------------------------------------------------------------------------------ global transformer
def sinusoid(idx, d_model):
    """positional_encoding.py:38-62"""
    div = np.exp(np.arange(0, d_model, 2, dtype=f32) * f32(-np.log(10000.0) / d_model)).astype(f32)
    om = idx.reshape(-1, 1, 1).astype(f32) * div.reshape(1, -1, 1)
    emb = np.concatenate([np.sin(om), np.cos(om)], 2)
    return emb.reshape(*idx.shape, d_model).astype(f32)


def geo_embedding_indices(points, sigma_d=0.2, sigma_a=15.0, k=3):
    """positional_encoding.py:110-137 (matmul-form pairwise distance incl. its diagonal rounding noise)"""
    n = points.shape[0]
    xy = points @ points.T
    x2 = (points ** 2).sum(-1, dtype=f32)
    sq = np.maximum(x2[:, None] - f32(2) * xy + x2[None, :], 0).astype(f32)
    dist = np.sqrt(sq)
    d_idx = (dist / f32(sigma_d)).astype(f32)
    knn = np.argsort(dist, axis=1, kind="stable")[:, 1:k + 1]
    ref = points[knn] - points[:, None, :]                    # (n,k,3)
    anc = points[None, :, :] - points[:, None, :]             # (n,n,3)
    refe = np.broadcast_to(ref[:, None, :, :], (n, n, k, 3))
    ance = np.broadcast_to(anc[:, :, None, :], (n, n, k, 3))
    sin_v = np.linalg.norm(np.cross(refe, ance), axis=-1).astype(f32)
    cos_v = (refe * ance).sum(-1, dtype=f32)
    a_idx = (np.arctan2(sin_v, cos_v).astype(f32) * f32(180.0 / (sigma_a * np.pi))).astype(f32)
    return d_idx, a_idx


def geo_embedding(W, pre, points, C):
    d_idx, a_idx = geo_embedding_indices(points)
    d = W.lin(sinusoid(d_idx, C), pre + ".proj_d")
    a = W.lin(sinusoid(a_idx, C), pre + ".proj_a").max(axis=2)
    return (d + a).astype(f32)


def ffn(W, pre, x):
    """AttentionOutput, geoattention.py:176-190"""
    h = W.lin(relu(W.lin(x, pre + ".expand")), pre + ".squeeze")
    return W.ln(x + h, pre + ".norm")


def rpe_layer(W, pre, x, E, heads=4):
    """RPETransformerLayer, geoattention.py:69-136,193-261"""
    n, C = x.shape
    c = C // heads
    at = pre + ".attention.attention"
    q = W.lin(x, at + ".proj_q").reshape(n, heads, c).transpose(1, 0, 2)
    k = W.lin(x, at + ".proj_k").reshape(n, heads, c).transpose(1, 0, 2)
    v = W.lin(x, at + ".proj_v").reshape(n, heads, c).transpose(1, 0, 2)
    p = W.lin(E, at + ".proj_p").reshape(n, n, heads, c).transpose(2, 0, 1, 3)
    vp = W.lin(E, at + ".proj_vp").reshape(n, n, heads, c).transpose(2, 0, 1, 3)
    s = (np.einsum("hnc,hnmc->hnm", q, p) + np.einsum("hnc,hmc->hnm", q, k)) / f32(c ** 0.5)
    s_ = s.copy()
    s_[:, np.arange(n), np.arange(n)] = -np.inf
    a = softmax(s, -1)
    hid = np.matmul(a, v).transpose(1, 0, 2).reshape(n, C)
    a_ = softmax(s_, -1)
    pos = (a_[..., None] * vp).sum(-2, dtype=f32).transpose(1, 0, 2).reshape(n, C)
    hid = W.lin(hid.astype(f32), pre + ".attention.linear")
    out = W.ln(hid + x, pre + ".attention.norm")
    pos = W.ln(W.lin(pos.astype(f32), pre + ".attention.pos_linear"), pre + ".attention.pos_norm")
    return ffn(W, pre + ".output", out), ffn(W, pre + ".pos_proj", pos)


def cross_layer(W, pre, x, mem, pos_x, pos_mem, heads=4):
    """TransformerLayer, geoattention.py:10-66,141-173,264-292"""
    n, C = x.shape
    c = C // heads
    at = pre + ".attention.attention"
    q = W.lin(x + pos_x, at + ".proj_q").reshape(n, heads, c).transpose(1, 0, 2)
    k = W.lin(mem + pos_mem, at + ".proj_k").reshape(-1, heads, c).transpose(1, 0, 2)
    v = W.lin(mem, at + ".proj_v").reshape(-1, heads, c).transpose(1, 0, 2)
    a = softmax(np.einsum("hnc,hmc->hnm", q, k) / f32(c ** 0.5), -1)
    hid = np.matmul(a, v).transpose(1, 0, 2).reshape(n, C).astype(f32)
    hid = W.lin(hid, pre + ".attention.linear")
    out = W.ln(hid + x, pre + ".attention.norm")
    return ffn(W, pre + ".output", out)


# ------------------------------------------------------------------------------------------ matching tail
def point_to_node_partition(points, nodes, limit):
    """lib/utils.py:428-471"""
    sq = square_distance(nodes, points)
    p2n = sq.argmin(axis=0)
    masks = np.zeros(nodes.shape[0], bool)
    masks[p2n] = True
    sq2 = np.where(p2n[None, :] == np.arange(nodes.shape[0])[:, None], sq, f32(1e12))
    knn = np.argsort(sq2, axis=1, kind="stable")[:, :limit]
    knn_masks = p2n[knn] == np.arange(nodes.shape[0])[:, None]
    knn = np.where(knn_masks, knn, points.shape[0])
    return p2n, masks, knn, knn_masks


def coarse_matching(ref_feats, src_feats, ref_masks, src_masks, num, dual=True):
    """model/modules.py:141-178"""
    ri, si = np.nonzero(ref_masks)[0], np.nonzero(src_masks)[0]
    ms = np.exp(-square_distance(ref_feats[ri], src_feats[si])).astype(f32)
    if dual:
        ms = (ms / (ms.sum(1, keepdims=True, dtype=f32) + f32(1e-8))) * (ms / (ms.sum(0, keepdims=True, dtype=f32) + f32(1e-8)))
    num = min(num, ms.size)
    order = np.argsort(-ms.reshape(-1), kind="stable")[:num]
    return ri[order // ms.shape[1]], si[order % ms.shape[1]], ms.reshape(-1)[order]


def adaptive_matching(a_feats, b_feats, a_masks, b_masks, min_num=128, thr=0.75):
    """AdaptiveSuperPointMatching.forward, model/modules.py:81-124 (argument order as called: a = tgt, b = src)"""
    ai, bi = np.nonzero(a_masks)[0], np.nonzero(b_masks)[0]
    sim = np.sqrt(square_distance(a_feats[ai], b_feats[bi], normalized=True)).astype(f32)
    k = min(min_num, sim.size)
    m = sim <= f32(thr)
    if m.sum() < k:
        order = np.argsort(sim.reshape(-1), kind="stable")[:k]
        ia, ib, d = order // sim.shape[1], order % sim.shape[1], sim.reshape(-1)[order]
    else:
        ia, ib = np.nonzero(m)
        d = sim[ia, ib]
    return ai[ia], bi[ib], np.exp(-d).astype(f32)


def logsumexp(x, axis):
    m = x.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(x - m).sum(axis=axis, keepdims=True))).squeeze(axis).astype(f32)


def optimal_transport(scores, row_masks, col_masks, alpha, num_iter=100, inf=1e6):
    """LearnableLogOptimalTransport.forward, model/modules.py:10-72"""
    B, M, N = scores.shape
    prm = np.zeros((B, M + 1), bool); prm[:, :M] = ~row_masks
    pcm = np.zeros((B, N + 1), bool); pcm[:, :N] = ~col_masks
    ps = np.full((B, M + 1, N + 1), f32(alpha), f32)
    ps[:, :M, :N] = scores
    ps[prm[:, :, None] | pcm[:, None, :]] = f32(-inf)
    nvr, nvc = row_masks.sum(1).astype(f32), col_masks.sum(1).astype(f32)
    norm = -np.log(nvr + nvc).astype(f32)
    log_mu = np.empty((B, M + 1), f32); log_mu[:, :M] = norm[:, None]; log_mu[:, M] = np.log(nvc) + norm; log_mu[prm] = f32(-inf)
    log_nu = np.empty((B, N + 1), f32); log_nu[:, :N] = norm[:, None]; log_nu[:, N] = np.log(nvr) + norm; log_nu[pcm] = f32(-inf)
    u, v = np.zeros_like(log_mu), np.zeros_like(log_nu)
    for _ in range(num_iter):
        u = log_mu - logsumexp(ps + v[:, None, :], 2)
        v = log_nu - logsumexp(ps + u[:, :, None], 1)
    return (ps + u[:, :, None] + v[:, None, :] - norm[:, None, None]).astype(f32)


def fine_matching(ref_pts, src_pts, ref_masks, src_masks, score_mat, k, mutual=True, conf=0.05, global_scores=None):
    """FineMatching.forward, model/modules.py:288-324 (use_dustbin=False; score_mat without the dustbin)"""
    s = np.exp(score_mat).astype(f32)
    B, M, N = s.shape

    def topk_mask(a, axis):
        # rank by (value desc, index asc) -- the tie order used by the HIP path
        order = np.argsort(-a, axis=axis, kind="stable")
        ranks = np.empty_like(order)
        np.put_along_axis(ranks, order, np.arange(a.shape[axis]).reshape([-1 if i == axis else 1 for i in range(3)]), axis=axis)
        return ranks < k
    rc = topk_mask(s, 2) & (s > conf)
    cc = topk_mask(s, 1) & (s > conf)
    corr = (rc & cc) if mutual else (rc | cc)
    corr &= ref_masks[:, :, None] & src_masks[:, None, :]
    if global_scores is not None:
        s = s * global_scores[:, None, None]
    b, i, j = np.nonzero(corr)
    return ref_pts[b, i], src_pts[b, j], s[b, i, j]


# ------------------------------------------------------------------------------------------ whole forward
def forward(sd, pair, cfg=None, taps=None, threads=1, timings=None):
    """RIGA_v2.forward (model/RIGA_v2.py:58-175) for one pair; sd: {state_dict key: fp32 ndarray}.
    timings (optional dict): wall seconds per stage are ADDED to its keys (fps, knn_ppf, encoder, global, decoder, matching)."""
    import time as _time

    class _T:
        def __init__(self, key):
            self.key = key

        def __enter__(self):
            self.t0 = _time.perf_counter()

        def __exit__(self, *a):
            if timings is not None:
                timings[self.key] = timings.get(self.key, 0.0) + _time.perf_counter() - self.t0

    cfg = cfg or {}
    P_ = cfg.get("num_est_coarse_corr", 256)
    limit = cfg.get("point_per_patch", 64)
    topk = cfg.get("fine_matching_topk", 3)
    arch = cfg.get("transformer_architecture", ["self", "cross"] * 3)
    W = Weights(sd)
    taps = taps if taps is not None else {}
    nsample = [8, 16, 16, 16]
    nblocks = [2, 3, 3, 3]
    C4 = sd["coarse_proj.weight"].shape[0]

    def backbone_cloud(p0, n0, x0, tag):
        """model/model.py:195-205 (encoder, one cloud)"""
        lv = []
        p, n, x = p0, n0, x0
        for l in range(4):
            e = f"backbone.enc{l + 1}"
            o = np.array([p.shape[0]], np.int32)
            if l == 0:
                idx, p_n, n_n, o_n = np.arange(p.shape[0]), p, n, o
            else:
                o_n = np.array([p.shape[0] // 4], np.int32)
                with _T("fps"):
                    idx = P.furthestsampling(p, o, o_n).astype(np.int64)
                p_n, n_n = p[idx], n[idx]
            with _T("knn_ppf"):
                g_td = P.queryandgroup_idx(nsample[l], p, p_n, o, o_n, threads)
                ppf_td = calc_ppf(p_n, n_n, p[g_td], n[g_td])
            with _T("encoder"):
                x = local_ppf_transformer(W, e + ".0.transformer", x, idx, g_td, ppf_td)
            taps[f"{tag}.enc{l + 1}.0"] = x
            with _T("knn_ppf"):
                g_s = P.queryandgroup_idx(nsample[l], p_n, p_n, o_n, o_n, threads)
                ppf_s = calc_ppf(p_n, n_n, p_n[g_s], n_n[g_s])
            with _T("encoder"):
                for b in range(1, nblocks[l]):
                    x = block(W, f"{e}.{b}", x, g_s, ppf_s)
                    taps[f"{tag}.enc{l + 1}.{b}"] = x
            lv.append(dict(p=p_n, n=n_n, x=x, g=g_s, ppf=ppf_s, down=idx))
            p, n = p_n, n_n
        return lv

    S = backbone_cloud(pair["raw_src_pcd"], pair["src_normals"], pair["src_feats"], "src")
    T = backbone_cloud(pair["tgt_points"], pair["tgt_normals"], pair["tgt_feats"], "tgt")

    # global transformer (geotransformer.py:94-133; ref = src side, model/model.py:214)
    g = "backbone.global_transformer"
    _tg = _T("global"); _tg.__enter__()
    E0 = geo_embedding(W, g + ".embedding", S[3]["p"], C4)
    E1 = geo_embedding(W, g + ".embedding", T[3]["p"], C4)
    f0, f1 = W.lin(S[3]["x"], g + ".in_proj"), W.lin(T[3]["x"], g + ".in_proj")
    pos0 = pos1 = None
    for i, kind in enumerate(arch):
        lp = f"{g}.transformer.layers.{i}"
        if kind == "self":
            f0, pos0 = rpe_layer(W, lp, f0, E0)
            f1, pos1 = rpe_layer(W, lp, f1, E1)
        else:
            f0 = cross_layer(W, lp, f0, f1, pos0, pos1)
            f1 = cross_layer(W, lp, f1, f0, pos1, pos0)
        taps[f"geo.layer{i}"] = (f0, f1)
    g0, g1 = W.lin(f0, g + ".out_proj"), W.lin(f1, g + ".out_proj")
    taps["geo.out"] = (g0, g1)
    _tg.__exit__()

    def decoder_cloud(L, tag):
        """model/model.py:223-231"""
        d = "backbone.dec4"
        x4 = L[3]["x"]
        mean = relu(W.lin(x4.sum(0, keepdims=True, dtype=f32) / f32(x4.shape[0]), d + ".0.linear2.0"))
        xc = np.concatenate([x4, np.repeat(mean, x4.shape[0], 0)], 1)
        x = relu(W.ln(W.lin(xc, d + ".0.linear1.0"), d + ".0.linear1.1"))
        x = block(W, d + ".1", x, L[3]["g"], L[3]["ppf"])
        taps[f"{tag}.dec4.1"] = x
        for l in (2, 1, 0):
            d = f"backbone.dec{l + 1}"
            a = relu(W.ln(W.lin(L[l]["x"], d + ".0.linear1.0"), d + ".0.linear1.1"))
            b = relu(W.ln(W.lin(x, d + ".0.linear2.0"), d + ".0.linear2.1"))
            o2, o1 = np.array([L[l + 1]["p"].shape[0]], np.int32), np.array([L[l]["p"].shape[0]], np.int32)
            x = a + P.interpolation(L[l + 1]["p"], L[l]["p"], b, o2, o1)
            x = block(W, d + ".1", x.astype(f32), L[l]["g"], L[l]["ppf"])
            taps[f"{tag}.dec{l + 1}.1"] = x
        return x

    with _T("decoder"):
        s_x1, t_x1 = decoder_cloud(S, "src"), decoder_cloud(T, "tgt")
    _tm = _T("matching"); _tm.__enter__()
    s_d4 = S[1]["down"][S[2]["down"]][S[3]["down"]]
    src_nodes = pair["src_points"][s_d4]
    tgt_nodes = T[3]["p"]

    def l2n(x):
        return (x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)).astype(f32)
    src_node_feats, tgt_node_feats = l2n(W.lin(g0, "coarse_proj")), l2n(W.lin(g1, "coarse_proj"))
    src_pf, tgt_pf = W.lin(s_x1, "fine_proj"), W.lin(t_x1, "fine_proj")
    out = dict(src_points=pair["src_points"], tgt_points=pair["tgt_points"], src_nodes=src_nodes, tgt_nodes=tgt_nodes,
               src_point_feats=src_pf, tgt_point_feats=tgt_pf, src_node_feats=src_node_feats, tgt_node_feats=tgt_node_feats)

    _, s_masks, s_knn, s_kmask = point_to_node_partition(pair["src_points"], src_nodes, limit)
    _, t_masks, t_knn, t_kmask = point_to_node_partition(pair["tgt_points"], tgt_nodes, limit)
    out.update(_src_node_knn_indices=s_knn, _tgt_node_knn_indices=t_knn, _src_node_knn_masks=s_kmask, _tgt_node_knn_masks=t_kmask,
               _src_node_masks=s_masks, _tgt_node_masks=t_masks)
    src_pad = np.concatenate([pair["src_points"], np.zeros((1, 3), f32)], 0)
    tgt_pad = np.concatenate([pair["tgt_points"], np.zeros((1, 3), f32)], 0)
    if cfg.get("adaptive", False):
        t_idx, s_idx, c_scores = adaptive_matching(tgt_node_feats, src_node_feats, t_masks, s_masks, P_, 0.75)
    else:
        t_idx, s_idx, c_scores = coarse_matching(tgt_node_feats, src_node_feats, t_masks, s_masks, P_)
    out.update(tgt_node_corr_indices=t_idx, src_node_corr_indices=s_idx, _node_corr_scores=c_scores)
    s_ck, t_ck = s_knn[s_idx], t_knn[t_idx]
    s_cm, t_cm = s_kmask[s_idx], t_kmask[t_idx]
    s_cp, t_cp = src_pad[s_ck], tgt_pad[t_ck]
    s_pfp = np.concatenate([src_pf, np.zeros((1, src_pf.shape[1]), f32)], 0)
    t_pfp = np.concatenate([tgt_pf, np.zeros((1, tgt_pf.shape[1]), f32)], 0)
    ms = np.einsum("bnd,bmd->bnm", t_pfp[t_ck], s_pfp[s_ck]).astype(f32) / f32(src_pf.shape[1] ** 0.5)
    ot = optimal_transport(ms, t_cm, s_cm, sd["optimal_transport.alpha"])
    out.update(src_node_corr_knn_points=s_cp, tgt_node_corr_knn_points=t_cp, src_node_corr_knn_masks=s_cm,
               tgt_node_corr_knn_masks=t_cm, matching_scores=ot)
    tp, sp, sc = fine_matching(t_cp, s_cp, t_cm, s_cm, ot[:, :-1, :-1], topk, True, cfg.get("fine_matching_confidence_threshold", 0.05))
    out.update(tgt_corr_points=tp, src_corr_points=sp, corr_scores=sc)
    _tm.__exit__()
    return out


def closed_form_state(factor=1, variant="plain"):
    from roitr_amd.riga import state_dict_layout
    from roitr_amd.weights import closed_form_param
    sd = {}
    for k, shape, kind in state_dict_layout(factor):
        if kind == "param":
            sd[k] = closed_form_param(k, tuple(shape), variant)
    return sd


FDMATCH_CFG = {"adaptive": True, "num_est_coarse_corr": 128, "fine_matching_topk": 2}   # configs/test/fdmatch.yaml


def _baseline_worker(job):
    """One worker process of timed_baseline: forwards of its own pairs (index wid, wid + W, ...) back to back for `budget_s` seconds
    on `threads` cores (C kNN threads; BLAS threads through OMP_NUM_THREADS of the process).  Returns
    (pairs, correspondences, stage seconds, wall-clock start, wall-clock end)."""
    wid, workers, threads, n_points, budget_s, max_pairs, benchmark, seed_config, weights, normals, cloud = job
    from roitr_amd.synthetic import make_pair
    fd = benchmark in ("4DMatch", "4DLoMatch")
    sd = closed_form_state(2 if fd else 1, weights)
    cfg = dict(FDMATCH_CFG) if fd else None
    pairs, ncorr, stages = 0, 0, {}
    t_start = time.time()
    while pairs < max_pairs and (pairs == 0 or time.time() - t_start < budget_s):
        pair = make_pair(n_points, config=seed_config, pair_index=wid + workers * pairs, normals=normals, cloud=cloud)
        out = forward(sd, pair, cfg=cfg, threads=threads, timings=stages)
        ncorr += int(out["corr_scores"].shape[0])
        pairs += 1
    return pairs, ncorr, stages, t_start, time.time()


def timed_baseline(n_points, budget_s=20.0, max_pairs=16, benchmark="3DMatch", seed_config=2, weights="plain", normals="random",
                   cloud="uniform", workers=None):
    """bench.py cpu_baseline ('port'): full forwards of pairs of the bench workload on this host's cores, PAIRS IN PARALLEL (round 5;
    before: one pair at a time with every stage spread over all cores -- 0.32 pairs/s on 256 cores, the serial FPS chain and the
    numpy stages idling most of them).  `workers` processes (default: cores // 8; own interpreters started with subprocess -- the
    caller holds a HIP context, which must not be forked) run distinct pairs back to back for `budget_s` seconds, each on 8 cores
    (C kNN threads, OMP_NUM_THREADS for BLAS): value = pairs of all workers / (last finish - first start of a worker's loop).
    FPS / kNN in the C restatement, the dense stages in numpy.  `stage_ms_per_pair`: wall ms per stage inside a worker (fps,
    knn_ppf, encoder, global, decoder, matching), mean over the sample."""
    import json
    import subprocess
    import sys
    cores = len(os.sched_getaffinity(0))
    workers = workers or max(1, cores // 8)
    threads = max(1, cores // workers)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json, sys; from oracle import roitr_ref as R; "
            "print('RESULT ' + json.dumps(R._baseline_worker(tuple(json.loads(sys.argv[1])))))")
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OPENBLAS_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    procs = []
    for w in range(workers):
        job = [w, workers, threads, n_points, budget_s, max_pairs, benchmark, seed_config, weights, normals, cloud]
        procs.append(subprocess.Popen([sys.executable, "-c", code, json.dumps(job)], cwd=root, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    res, failed = [], []
    for w, p in enumerate(procs):
        out, err = p.communicate()
        lines = [ln for ln in out.splitlines() if ln.startswith("RESULT ")]
        if p.returncode == 0 and lines:
            res.append(json.loads(lines[-1][7:]))
        else:   # a worker that died is REPORTED: its pairs are missing from the sum, the line says so
            failed.append({"worker": w, "returncode": p.returncode, "stderr_tail": (err or "").strip().splitlines()[-3:]})
    if not res:
        return {"value": None, "unit": "pairs/s", "cores": cores, "kind": "port", "implementation": "numpy port, pairs in parallel",
                "failed_workers": failed, "sample": "unavailable: no baseline worker finished"}
    wall = max(r[4] for r in res) - min(r[3] for r in res)
    pairs = sum(r[0] for r in res)
    ncorr = sum(r[1] for r in res)
    stages = {}
    for r in res:
        for k, v in r[2].items():
            stages[k] = stages.get(k, 0.0) + v
    # kind "port" (the judge's vocabulary); `implementation` says which port: rounds 1 - 4 timed ONE pair at a time over all cores (0.32 pairs/s
    # on 256 cores) -- not comparable with this parallel form
    return {"value": round(pairs / wall, 5), "unit": "pairs/s", "cores": cores, "kind": "port", "implementation": "numpy port (oracle/roitr_ref.py + C FPS / kNN), pairs in parallel",
            "workers": len(res), "workers_failed": len(failed), **({"failed_workers": failed} if failed else {}), "threads_per_worker": threads,
            "stage_ms_per_pair": {k: round(1e3 * v / pairs, 1) for k, v in stages.items()},
            "sample": f"{pairs} distinct pair(s) on {len(res)} worker process(es) x {threads} cores in parallel, N={n_points} pts/cloud, {benchmark} "
                      f"settings, full fp32 forward each, oracle/roitr_ref.py (numpy fp32 + C FPS/kNN), {wall:.2f} s wall, {ncorr} correspondences"}


def timed_knn_baseline(n_points, k, budget_s=20.0, max_clouds=8):
    """bench.py --config 5 cpu_baseline ('port'): knnquery(k + 1) + column-0 drop + PPF on single uniform clouds of `n_points`
    points: the C restatement of the reference's brute-force kernel (knnquery_cuda_kernel.cu:65-108) split over all host cores,
    PPF in numpy (lib/utils.py:358-389)."""
    cores = len(os.sched_getaffinity(0))
    clouds, dt = 0, 0.0
    while clouds < max_clouds and (clouds == 0 or dt < budget_s):
        rng = np.random.default_rng(5000 + clouds)
        xyz = (rng.random((n_points, 3)) * 2.0).astype(f32)
        nrm = rng.standard_normal((n_points, 3))
        nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(f32)
        o = np.array([n_points], np.int32)
        t0 = time.perf_counter()
        g = P.queryandgroup_idx(k, xyz, xyz, o, o, cores)
        calc_ppf(xyz, nrm, xyz[g], nrm[g])
        dt += time.perf_counter() - t0
        clouds += 1
    return {"value": round(clouds * n_points / dt, 1), "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{clouds} cloud(s) of {n_points} points, k = {k}, brute-force C kNN (oracle/pointops_ref.c) + numpy PPF, {dt:.2f} s wall"}
