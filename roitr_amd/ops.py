"""Operator-level mirrors of the reference's lib/utils.py hot-path helpers, backed by HIP kernels.

Each function cites the reference function it stands for.  ROCm device tensors only.
"""
import torch

from . import _lib as L


def _i32c(t):
    return t.to(torch.int32).contiguous()


def calc_ppf(points, point_normals, ref_points, ref_normals, group_idx):
    """lib/utils.py:358-389 calc_ppf_gpu(points, point_normals, ref_points[group_idx], ref_normals[group_idx]).

    The reference takes pre-gathered (m,k,3) patches; gathering is fused here, so the caller passes the
    un-gathered reference cloud and the (m,k) indices instead.  Returns (m,k,4) float32."""
    m, k = group_idx.shape
    out = torch.empty((m, k, 4), dtype=torch.float32, device=points.device)
    grp = _i32c(group_idx)
    L.check(L.lib().roitr_calc_ppf(m, k, L.ptr(points.contiguous()), L.ptr(point_normals.contiguous()),
                                   L.ptr(ref_points.contiguous()), L.ptr(ref_normals.contiguous()), L.ptr(grp), L.ptr(out),
                                   L.stream_ptr()), "calc_ppf")
    return out


def calc_ppf_gpu(points, point_normals, patches, patch_normals):
    """Signature-compatible with lib/utils.py:358 (pre-gathered patches (m,k,3))."""
    m, k, _ = patches.shape
    grp = torch.arange(m * k, dtype=torch.int32, device=points.device).view(m, k)
    return calc_ppf(points, point_normals, patches.reshape(-1, 3), patch_normals.reshape(-1, 3), grp)


# ------------------------------------------------------------------------------------------------
# Matching-tail operators (single pair / single batch of patches), thin ctypes fronts of the batched kernels.
# ------------------------------------------------------------------------------------------------
import ctypes  # noqa: E402

_P = ctypes.c_void_p


class _Coarse(ctypes.Structure):
    _fields_ = [("pairs", ctypes.c_int), ("C", ctypes.c_int), ("num_corr", ctypes.c_int), ("dual_norm", ctypes.c_int),
                ("max_ref", ctypes.c_int), ("max_src", ctypes.c_int), ("feats", _P), ("node_offset", _P), ("node_masks", _P),
                ("scratch", _P), ("scratch_stride", ctypes.c_long), ("tgt_corr", _P), ("src_corr", _P), ("corr_scores", _P),
                ("n_corr", _P), ("xy", _P), ("xy_stride", ctypes.c_long), ("xy_ld", ctypes.c_int), ("lds_cap", ctypes.c_int)]


class _OT(ctypes.Structure):
    _fields_ = [("pairs", ctypes.c_int), ("num_corr", ctypes.c_int), ("limit", ctypes.c_int), ("num_iter", ctypes.c_int),
                ("n_corr", _P), ("scores", _P), ("row_masks", _P), ("col_masks", _P), ("alpha", _P), ("out", _P),
                ("pair_off", _P), ("slots", ctypes.c_int)]


class _Fine(ctypes.Structure):
    _fields_ = [("pairs", ctypes.c_int), ("num_corr", ctypes.c_int), ("limit", ctypes.c_int), ("k", ctypes.c_int),
                ("mutual", ctypes.c_int), ("conf", ctypes.c_float), ("n_corr", _P), ("ot", _P), ("row_masks", _P),
                ("col_masks", _P), ("row_pts", _P), ("col_pts", _P), ("global_scores", _P), ("flags", _P), ("counts", _P),
                ("offsets", _P), ("n_out", _P), ("out_row_pts", _P), ("out_col_pts", _P), ("out_scores", _P), ("out_patch", _P), ("out_cap", ctypes.c_long),
                ("pair_off", _P), ("slots", ctypes.c_int), ("pair_starts", _P)]


def point_to_node_partition(points, nodes, point_limit):
    """lib/utils.py:428-471 -> (point_to_node (N,) i64, node_masks (M,) bool, node_knn_indices (M,K) i64, node_knn_masks bool)"""
    dev = points.device
    N, M = points.shape[0], nodes.shape[0]
    i32 = torch.int32
    po = torch.tensor([N], dtype=i32, device=dev)
    no = torch.tensor([M], dtype=i32, device=dev)
    con = torch.zeros(M, dtype=i32, device=dev)
    p2n = torch.zeros(N, dtype=i32, device=dev)
    p2nd = torch.zeros(N, dtype=torch.float32, device=dev)
    nm = torch.zeros(M, dtype=i32, device=dev)
    kidx = torch.zeros((M, point_limit), dtype=i32, device=dev)
    kmask = torch.zeros((M, point_limit), dtype=i32, device=dev)
    L.check(L.lib().roitr_point_to_node_partition(1, N, M, L.ptr(points.contiguous()), L.ptr(po), L.ptr(nodes.contiguous()), L.ptr(no),
                                                  L.ptr(con), int(point_limit), L.ptr(p2n), L.ptr(p2nd), L.ptr(nm), L.ptr(kidx),
                                                  L.ptr(kmask), L.stream_ptr()), "point_to_node_partition")
    return p2n.long(), nm.bool(), kidx.long(), kmask.bool()


def coarse_matching(ref_feats, src_feats, ref_masks, src_masks, num_correspondences=256, dual_normalization=True):
    """model/modules.py:141-178 CoarseMatching.forward -> (ref_corr_indices, src_corr_indices, corr_scores)"""
    dev = ref_feats.device
    nr, ns = ref_feats.shape[0], src_feats.shape[0]
    lib = L.lib()
    lib.roitr_coarse_scratch_floats.restype = ctypes.c_size_t
    feats = torch.cat([src_feats, ref_feats], 0).contiguous().float()   # cloud order: [src, tgt(=ref)]
    masks = torch.cat([src_masks, ref_masks], 0).to(torch.int32).contiguous()
    off = torch.tensor([ns, ns + nr], dtype=torch.int32, device=dev)
    stride = lib.roitr_coarse_scratch_floats(nr, ns)
    scratch = torch.empty(stride, dtype=torch.float32, device=dev)
    P = int(num_correspondences)
    tc = torch.zeros(P, dtype=torch.int32, device=dev)
    sc = torch.zeros(P, dtype=torch.int32, device=dev)
    cs = torch.zeros(P, dtype=torch.float32, device=dev)
    nc = torch.zeros(1, dtype=torch.int32, device=dev)
    a = _Coarse(1, feats.shape[1], P, int(dual_normalization), nr, ns, L.ptr(feats), L.ptr(off), L.ptr(masks), L.ptr(scratch),
                stride, L.ptr(tc), L.ptr(sc), L.ptr(cs), L.ptr(nc), L.ptr(None), 0, 0)
    L.check(lib.roitr_coarse_matching(ctypes.byref(a), L.stream_ptr()), "coarse_matching")
    n = int(nc.item())
    return tc[:n].long(), sc[:n].long(), cs[:n]


def optimal_transport(scores, row_masks, col_masks, alpha, num_iter=100):
    """model/modules.py:28-68 LearnableLogOptimalTransport.forward: (B,64,64) -> (B,65,65)"""
    dev = scores.device
    B, M, N = scores.shape
    out = torch.zeros((B, M + 1, N + 1), dtype=torch.float32, device=dev)
    nc = torch.tensor([B], dtype=torch.int32, device=dev)
    al = torch.as_tensor(alpha, dtype=torch.float32, device=dev).reshape(1).contiguous()
    sc = scores.contiguous().float()
    rm = row_masks.to(torch.int32).contiguous()   # named: temporaries must outlive the launch
    cm = col_masks.to(torch.int32).contiguous()
    a = _OT(1, B, M, int(num_iter), L.ptr(nc), L.ptr(sc), L.ptr(rm), L.ptr(cm), L.ptr(al), L.ptr(out))
    L.check(L.lib().roitr_optimal_transport(ctypes.byref(a), L.stream_ptr()), "optimal_transport")
    return out


def fine_matching(ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, matching_scores, k, mutual=True,
                  confidence_threshold=0.05, global_scores=None):
    """model/modules.py:288-324 FineMatching.forward (use_dustbin=False).  matching_scores: the (B,65,65) OT output
    (its dustbin row/column are dropped inside, RIGA_v2.py:159-160) -> (ref_corr_points, src_corr_points, corr_scores)"""
    dev = matching_scores.device
    B, Lp1, _ = matching_scores.shape
    Lm = Lp1 - 1
    i32 = torch.int32
    cap = B * Lm * Lm
    nc = torch.tensor([B], dtype=i32, device=dev)
    flags = torch.zeros(B * Lm * Lm, dtype=torch.uint8, device=dev)
    counts = torch.zeros(B, dtype=i32, device=dev)
    offs = torch.zeros(B, dtype=i32, device=dev)
    n_out = torch.zeros(1, dtype=i32, device=dev)
    o_r = torch.zeros((cap, 3), dtype=torch.float32, device=dev)
    o_c = torch.zeros((cap, 3), dtype=torch.float32, device=dev)
    o_s = torch.zeros(cap, dtype=torch.float32, device=dev)
    rm, cm = ref_knn_masks.to(i32).contiguous(), src_knn_masks.to(i32).contiguous()
    rp, cp = ref_knn_points.contiguous().float(), src_knn_points.contiguous().float()
    ms = matching_scores.contiguous().float()
    gs = global_scores.contiguous().float() if global_scores is not None else None
    a = _Fine(1, B, Lm, int(k), int(mutual), float(confidence_threshold), L.ptr(nc), L.ptr(ms), L.ptr(rm), L.ptr(cm), L.ptr(rp),
              L.ptr(cp), L.ptr(gs), L.ptr(flags), L.ptr(counts), L.ptr(offs), L.ptr(n_out), L.ptr(o_r), L.ptr(o_c), L.ptr(o_s), L.ptr(None), cap)
    L.check(L.lib().roitr_fine_matching(ctypes.byref(a), L.stream_ptr()), "fine_matching")
    n = int(n_out.item())
    return o_r[:n], o_c[:n], o_s[:n]


class _Gemm(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("A", _P), ("A2", _P), ("lda", ctypes.c_int),
                ("a_idx", _P), ("a_limit", ctypes.c_int), ("W", _P), ("ldw", ctypes.c_int), ("w_idx", _P), ("w_limit", ctypes.c_int),
                ("bias", _P), ("alpha", ctypes.c_float), ("relu", ctypes.c_int), ("C", _P), ("ldc", ctypes.c_int),
                ("batch", ctypes.c_int), ("sA", ctypes.c_long), ("sW", ctypes.c_long), ("sC", ctypes.c_long),
                ("sBias", ctypes.c_long), ("sAidx", ctypes.c_long), ("sWidx", ctypes.c_long), ("seg_off", _P),
                ("seg_a0", ctypes.c_int), ("seg_w0", ctypes.c_int),
                ("ln_gamma", _P), ("ln_beta", _P), ("ln_res", _P), ("ln_res_idx", _P), ("ln_post", _P),
                ("ln_relu", ctypes.c_int), ("ln_eps", ctypes.c_float), ("bf16", ctypes.c_int),
                ("A_cat", _P), ("lda_cat", ctypes.c_int), ("k_cat", ctypes.c_int),
                ("ip_feat", _P), ("ip_idx", _P), ("ip_dist2", _P), ("a_cat_idx", _P), ("batch_live", _P), ("w_piece", ctypes.c_long)]


BF16_W, BF16_A, BF16_C, BF16_X3 = 1, 2, 4, 8   # RoitrGemm::bf16 flags (include/roitr_engine.h)


def split_bf16x3(weight):
    """roitr_split_bf16x3: (N, K) fp32 -> (3, N, K) bf16 with weight == pieces.float().sum(0) exactly (csrc/gemm_x3.hip)."""
    w = weight.contiguous().float()
    out = torch.empty((3,) + tuple(w.shape), dtype=torch.bfloat16, device=w.device)
    L.check(L.lib().roitr_split_bf16x3(ctypes.c_long(w.numel()), L.ptr(w), L.ptr(out), ctypes.c_long(w.numel()), L.stream_ptr()), "split_bf16x3")
    return out


def _bf16_flags(bf16, x, weight, out_bf16):
    """bf16 operand mode of the GEMM: the weight is stored bf16, x is rounded while staged unless it already is a bfloat16
    tensor, the output is stored bf16 on request.  Returns (x, weight, flags)."""
    if not bf16:
        return x.contiguous().float(), weight.contiguous().float(), 0
    flags = BF16_W | (BF16_C if out_bf16 else 0)
    w = weight.contiguous().to(torch.bfloat16)
    if x.dtype == torch.bfloat16:
        return x.contiguous(), w, flags | BF16_A
    return x.contiguous().float(), w, flags


def _k_cat(g, x, x_cat, keep, x_cat_idx=None, addend=None):
    """RoitrGemm::A_cat: the product reads [x | x_cat] along K without the concatenation ever existing (fp32 fast path only).
    x_cat_idx: row gather of x_cat alone (RoitrGemm::a_cat_idx); addend: RoitrGemm::A2, added to the x part."""
    if addend is not None:
        ad = addend.contiguous().float()
        keep.append(ad)
        g.A2 = L.ptr(ad)
    if x_cat is None:
        return
    xc = x_cat.contiguous().float()
    keep.append(xc)
    g.A_cat, g.lda_cat, g.k_cat = L.ptr(xc), xc.shape[1], x.shape[1]
    g.K = x.shape[1] + xc.shape[1]
    if x_cat_idx is not None:
        ix = x_cat_idx.contiguous().to(torch.int32)
        keep.append(ix)
        g.a_cat_idx = L.ptr(ix)


def _x3(g, weight, keep):
    """RoitrGemm::bf16 = ROITR_BF16_X3: the weight as its three bf16 pieces (csrc/gemm_x3.hip)."""
    w3 = split_bf16x3(weight)
    keep.append(w3)
    g.W, g.bf16, g.w_piece = L.ptr(w3), BF16_X3, weight.numel()


def linear(x, weight, bias=None, relu=False, alpha=1.0, bf16=False, out_bf16=False, x_cat=None, x_cat_idx=None, addend=None, x3=False):
    """act(alpha * x @ weight.T + bias): the kernel behind every nn.Linear of the path.  Default: the fp32 MFMA GEMM.
    bf16=True: the bf16-operand kernel (csrc/gemm_bf16.hip; weights stored bf16, fp32 accumulate); x may be a bfloat16
    tensor (stored-bf16 activation), out_bf16 stores the result in bf16.  x_cat: a second operand block, the product is
    [x | x_cat] @ weight.T (the K-concatenated A of the folded block transformers)."""
    x, weight, flags = _bf16_flags(bf16, x, weight, out_bf16)
    M, K = x.shape
    N = weight.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16 if flags & BF16_C else torch.float32, device=x.device)
    b = bias.contiguous().float() if bias is not None else None
    g = _Gemm(M, N, K, L.ptr(x), L.ptr(None), K, L.ptr(None), 0, L.ptr(weight), K, L.ptr(None), 0, L.ptr(b), float(alpha), int(relu),
              L.ptr(out), N, 1, 0, 0, 0, 0, 0, 0, L.ptr(None), 0, 0)
    g.bf16 = flags
    keep = []
    _k_cat(g, x, x_cat, keep, x_cat_idx, addend)
    g.ldw = weight.shape[1]
    if x3:   # fp32 in, fp32 out, products on the bf16 matrix cores by the three-way split (csrc/gemm_x3.hip)
        _x3(g, weight, keep)
    L.check(L.lib().roitr_gemm(ctypes.byref(g), L.stream_ptr()), "gemm")
    return out


def linear_layernorm(x, weight, bias, gamma, beta, res=None, res_idx=None, post=None, relu=False, eps=1e-5, bf16=False, out_bf16=False,
                     x_cat=None, interp=None, x_cat_idx=None, addend=None, x3=False):
    """[relu](LayerNorm(x @ weight.T + bias + res[res_idx]) * gamma + beta + post) in ONE launch (64 / 128 / 256 output
    channels): the nn.Linear -> (+ residual) -> nn.LayerNorm call sites of attention.py:319, model/model.py:89-97,138-140.
    bf16 / out_bf16 as in linear().  interp = (feat (R, N), idx (M, 3) int32, dist2 (M, 3)): TransitionUp's three-nearest-neighbour
    interpolation (pointops.py:168-182) of the rows of `feat`, added after the activation (fp32 kernel only)."""
    x, weight, flags = _bf16_flags(bf16, x, weight, out_bf16)
    M, K = x.shape
    N = weight.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16 if flags & BF16_C else torch.float32, device=x.device)
    c = lambda t, dt=torch.float32: None if t is None else t.contiguous().to(dt)
    b, gm, bt, rs, ri, po = c(bias), c(gamma), c(beta), c(res), c(res_idx, torch.int32), c(post)
    g = _Gemm(M, N, K, L.ptr(x), L.ptr(None), K, L.ptr(None), 0, L.ptr(weight), K, L.ptr(None), 0, L.ptr(b), 1.0, 0,
              L.ptr(out), N, 1, 0, 0, 0, 0, 0, 0, L.ptr(None), 0, 0, L.ptr(gm), L.ptr(bt), L.ptr(rs), L.ptr(ri), L.ptr(po), int(relu), float(eps))
    g.bf16 = flags
    keep = []
    _k_cat(g, x, x_cat, keep, x_cat_idx, addend)
    g.ldw = weight.shape[1]
    if interp is not None:
        keep += [c(interp[0]), c(interp[1], torch.int32), c(interp[2])]
        g.ip_feat, g.ip_idx, g.ip_dist2 = L.ptr(keep[-3]), L.ptr(keep[-2]), L.ptr(keep[-1])
    if x3:
        _x3(g, weight, keep)
    L.check(L.lib().roitr_gemm(ctypes.byref(g), L.stream_ptr()), "gemm+layernorm")
    return out


def add_layernorm_interp(x, gamma, beta, feat, idx, dist2, res=None, res_idx=None, relu=False, eps=1e-5):
    """[relu](LayerNorm(x + res[res_idx]) * gamma + beta) + three-nearest-neighbour interpolation of the rows of `feat`
    (roitr_add_layernorm_interp: the two-launch twin of linear_layernorm(..., interp=...))."""
    c = lambda t, dt=torch.float32: None if t is None else t.contiguous().to(dt)
    x, gm, bt, rs, ri, ft, ix, d2 = c(x), c(gamma), c(beta), c(res), c(res_idx, torch.int32), c(feat), c(idx, torch.int32), c(dist2)
    M, C = x.shape
    out = torch.empty_like(x)
    L.check(L.lib().roitr_add_layernorm_interp(M, C, L.ptr(x), L.ptr(rs), L.ptr(ri), L.ptr(gm), L.ptr(bt), int(relu), ctypes.c_float(eps),
                                               L.ptr(ft), L.ptr(ix), L.ptr(d2), L.ptr(out), L.stream_ptr()), "add_layernorm_interp")
    return out


def adaptive_superpoint_matching(src_feats, tgt_feats, src_masks, tgt_masks, min_num_correspondences=128, similarity_threshold=0.75):
    """model/modules.py:81-124 AdaptiveSuperPointMatching.forward (argument names as in the reference: the FIRST set
    indexes the first returned index list).  Returns (src_corr_indices, tgt_corr_indices, corr_scores)."""
    dev = src_feats.device
    na, nb = src_feats.shape[0], tgt_feats.shape[0]
    lib = L.lib()
    lib.roitr_coarse_scratch_floats.restype = ctypes.c_size_t
    xy = linear(src_feats, tgt_feats)                       # (na, nb) feature dot products
    feats = torch.cat([tgt_feats, src_feats], 0).contiguous().float()      # engine cloud order: [second set, first set]
    masks = torch.cat([tgt_masks, src_masks], 0).to(torch.int32).contiguous()
    off = torch.tensor([nb, nb + na], dtype=torch.int32, device=dev)
    stride = lib.roitr_coarse_scratch_floats(na, nb)
    scratch = torch.empty(stride, dtype=torch.float32, device=dev)
    cap = na * nb
    ia = torch.zeros(cap, dtype=torch.int32, device=dev)
    ib = torch.zeros(cap, dtype=torch.int32, device=dev)
    sc = torch.zeros(cap, dtype=torch.float32, device=dev)
    nc = torch.zeros(1, dtype=torch.int32, device=dev)
    a = _Coarse(1, feats.shape[1], cap, 0, na, nb, L.ptr(feats), L.ptr(off), L.ptr(masks), L.ptr(scratch), stride, L.ptr(ia), L.ptr(ib),
                L.ptr(sc), L.ptr(nc), L.ptr(xy), 0, nb)
    L.check(lib.roitr_adaptive_matching(ctypes.byref(a), int(min_num_correspondences), ctypes.c_float(similarity_threshold),
                                        L.stream_ptr()), "adaptive_matching")
    n = int(nc.item())
    return ia[:n].long(), ib[:n].long(), sc[:n]


def geo_table_build(div_term, w_d, b_d, w_a, b_a, interval=2.0, d_range=48.0, a_range=12.0):
    """Host fit of the function table of the geometric embedding (csrc/geo_table.hip): returns (table on the weights' device,
    n_int_d, n_int_a, fit) with fit = [fit error d, amplitude d, fit error a, amplitude a, largest per-channel relative error d, a]
    measured in float64 on 64 probe points per interval."""
    import math
    C = int(w_d.shape[0])
    nd, na = int(math.ceil(d_range / interval)), int(math.floor(a_range / interval)) + 1
    lib = L.lib()
    h = lambda t: t.detach().float().cpu().contiguous()
    div_h, wd_h, bd_h, wa_h, ba_h = map(h, (div_term, w_d, b_d, w_a, b_a))
    table = torch.empty(int(lib.roitr_geo_table_floats(C, nd, na)), dtype=torch.float32)
    fit = (ctypes.c_double * 6)()
    hp = L.host_ptr
    L.check(lib.roitr_geo_table_build(C, hp(div_h), hp(wd_h), hp(bd_h), hp(wa_h), hp(ba_h), L.c_float(interval), nd, na,
                                      hp(table), fit), "geo_table_build")
    return table.to(w_d.device), nd, na, list(fit)


def geo_embed_table(d_idx, a_idx, table, interval, n_int_d, n_int_a, div_term, w_d, b_d, w_a, b_a, out_bf16=False):
    """The embedding of geo_embed() evaluated from the function table; values outside the table are evaluated directly."""
    rows, k = int(a_idx.shape[0]), int(a_idx.shape[1])
    C = int(w_d.shape[0])
    f = lambda t: t.contiguous().float()
    d_idx, a_idx, table, div_term, w_d, b_d, w_a, b_a = map(f, (d_idx, a_idx, table, div_term, w_d, b_d, w_a, b_a))
    out = torch.empty((rows, C), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=d_idx.device)
    L.check(L.lib().roitr_geo_embed_table(ctypes.c_long(rows), C, k, L.ptr(d_idx), L.ptr(a_idx), L.ptr(table), L.c_float(interval),
                                          int(n_int_d), int(n_int_a), L.ptr(div_term), L.ptr(w_d), L.ptr(b_d), L.ptr(w_a), L.ptr(b_a),
                                          L.ptr(out), 1 if out_bf16 else 0, L.stream_ptr()), "geo_embed_table")
    return out


def geo_embed(d_idx, a_idx, div_term, w_d, b_d, w_a, b_a, split=False, bf16=False):
    """positional_encoding.py:139-154 fused: proj_d(sinusoid(d_idx)) + max_k proj_a(sinusoid(a_idx[:, k])) for `rows` index
    rows.  split=True: the opt-in three-way bf16 split on the bf16 matrix cores (fp32-level accuracy); bf16=True: plain bf16
    operands (weights stored bf16, the sinusoid rounded to bf16, fp32 accumulate) -- the engine's bf16 operand mode."""
    rows, k = int(a_idx.shape[0]), int(a_idx.shape[1])
    C = int(w_d.shape[0])
    f = lambda t: t.contiguous().float()
    d_idx, a_idx, div_term, w_d, b_d, w_a, b_a = map(f, (d_idx, a_idx, div_term, w_d, b_d, w_a, b_a))
    out = torch.empty((rows, C), dtype=torch.float32, device=d_idx.device)
    lib = L.lib()
    if bf16:
        wdh, wah = w_d.to(torch.bfloat16).contiguous(), w_a.to(torch.bfloat16).contiguous()
        L.check(lib.roitr_geo_embed_bf16(ctypes.c_long(rows), C, k, L.ptr(d_idx), L.ptr(a_idx), L.ptr(div_term), L.ptr(wdh), L.ptr(b_d),
                                         L.ptr(wah), L.ptr(b_a), L.ptr(out), L.stream_ptr()), "geo_embed_bf16")
        return out
    if split:
        wd3 = torch.empty((3, C, C), dtype=torch.int16, device=out.device)
        wa3 = torch.empty((3, C, C), dtype=torch.int16, device=out.device)
        L.check(lib.roitr_split3_bf16(ctypes.c_long(C * C), L.ptr(w_d), L.ptr(wd3), L.stream_ptr()), "split3")
        L.check(lib.roitr_split3_bf16(ctypes.c_long(C * C), L.ptr(w_a), L.ptr(wa3), L.stream_ptr()), "split3")
        L.check(lib.roitr_geo_embed_split(ctypes.c_long(rows), C, k, L.ptr(d_idx), L.ptr(a_idx), L.ptr(div_term), L.ptr(wd3), L.ptr(b_d),
                                          L.ptr(wa3), L.ptr(b_a), L.ptr(out), L.stream_ptr()), "geo_embed_split")
    else:
        L.check(lib.roitr_geo_embed(ctypes.c_long(rows), C, k, L.ptr(d_idx), L.ptr(a_idx), L.ptr(div_term), L.ptr(w_d), L.ptr(b_d),
                                    L.ptr(w_a), L.ptr(b_a), L.ptr(out), L.stream_ptr()), "geo_embed")
    return out


class _LocalAttnFold(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int), ("in_dim", ctypes.c_int), ("H", ctypes.c_int),
                ("x", ctypes.c_void_p), ("ldx", ctypes.c_int), ("q", ctypes.c_void_p), ("ldq", ctypes.c_int),
                ("qt", ctypes.c_void_p), ("group_idx", ctypes.c_void_p), ("ppf", ctypes.c_void_p),
                ("wpe", ctypes.c_void_p), ("wvpe", ctypes.c_void_p), ("bvpe", ctypes.c_void_p), ("scale", ctypes.c_float),
                ("xbar", ctypes.c_void_p), ("vpart", ctypes.c_void_p), ("node_order", ctypes.c_void_p), ("ldqt", ctypes.c_int)]


def local_attention_fold(x, q, qt, group_idx, ppf, wpe, wvpe, bvpe, node_order=None, packed=False):
    """TransitionDown form of the local PPF attention with the key / value projections folded into the query side
    (csrc/local_attn.hip local_attn_fold_kernel; include/roitr_engine.h RoitrLocalAttnFold).  x (N_in, I) input rows, q (M, H),
    qt (M, 4, I) = Wk'_h^T q_h per head, group_idx (M, 16) int32 rows of x, ppf (M, 16, 4), wpe / wvpe (H, 4), bvpe (H).
    Returns (xbar (M, 4, I) = sum_j a_hj x_j, vpart (M, H) = Wvpe_h pbar_h + bvpe_h): the attention output of the unfolded form is
    vpart + [Wv'_h xbar_h + bv'_h]_h.  packed=True: q and qt are handed over as the two column blocks of ONE (M, H + 4 I) buffer (row
    stride H + 4 I for both: RoitrLocalAttnFold.ldq / ldqt), the way the engine's single q | q~ GEMM leaves them."""
    f = lambda t: t.contiguous().float()
    x, q, qt, ppf, wpe, wvpe, bvpe = f(x), f(q), f(qt), f(ppf), f(wpe), f(wvpe), f(bvpe)
    both = torch.cat([q, qt.reshape(q.shape[0], -1)], 1).contiguous() if packed else None
    group_idx = _i32c(group_idx)
    M, H, I = int(q.shape[0]), int(q.shape[1]), int(x.shape[1])
    if int(group_idx.shape[1]) != 16:
        raise L.RoitrError("local_attention_fold: 16 neighbours per node")
    xbar = torch.empty((M, 4, I), dtype=torch.float32, device=x.device)
    vpart = torch.empty((M, H), dtype=torch.float32, device=x.device)
    a = _LocalAttnFold()
    a.M, a.in_dim, a.H = M, I, H
    a.x, a.ldx, a.q, a.ldq, a.qt = L.ptr(x), I, L.ptr(q), H, L.ptr(qt)
    if packed:
        a.q, a.ldq, a.ldqt = L.ptr(both), H + 4 * I, H + 4 * I
        a.qt = ctypes.c_void_p(both.data_ptr() + 4 * H)
    a.group_idx, a.ppf, a.wpe, a.wvpe, a.bvpe = L.ptr(group_idx), L.ptr(ppf), L.ptr(wpe), L.ptr(wvpe), L.ptr(bvpe)
    a.scale, a.xbar, a.vpart = 1.0 / float(H // 4) ** 0.5, L.ptr(xbar), L.ptr(vpart)
    no = f(node_order) if node_order is not None else None
    a.node_order = L.ptr(no)
    L.check(L.lib().roitr_local_attention_fold(ctypes.byref(a), L.stream_ptr()), "local_attention_fold")
    return xbar, vpart


class _LocalTd(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int), ("in_dim", ctypes.c_int), ("H", ctypes.c_int),
                ("x", ctypes.c_void_p), ("node_idx", ctypes.c_void_p), ("group_idx", ctypes.c_void_p), ("ppf", ctypes.c_void_p),
                ("node_order", ctypes.c_void_p), ("wqqt", ctypes.c_void_p), ("bqqt", ctypes.c_void_p), ("wv", ctypes.c_void_p),
                ("bv", ctypes.c_void_p), ("wpe", ctypes.c_void_p), ("wvpe", ctypes.c_void_p), ("bvpe", ctypes.c_void_p),
                ("wcat", ctypes.c_void_p), ("bcat", ctypes.c_void_p), ("norm_w", ctypes.c_void_p), ("norm_b", ctypes.c_void_p),
                ("wout", ctypes.c_void_p), ("bout", ctypes.c_void_p), ("scale", ctypes.c_float), ("eps", ctypes.c_float), ("out", ctypes.c_void_p)]


def local_td(x, node_idx, group_idx, ppf, w, node_order=None):
    """The TransitionDown transformer of the 64 -> 128 wide level in one launch (csrc/local_block.hip local_td_kernel;
    include/roitr_engine.h RoitrLocalTd).  x (N_in, 64) input rows, node_idx (M,) int32 row of x of every node, group_idx (M, 16) int32
    rows of x, ppf (M, 16, 4); w: dict of the folded weights wqqt (384, 64), bqqt (384), wv (128, 64), bv (128), wpe / wvpe (128, 4),
    bvpe (128), wcat (128, 192), bcat, norm_w, norm_b, wout (128, 128), bout.  Returns (M, 128)."""
    f = lambda t: t.contiguous().float()
    x, ppf = f(x), f(ppf)
    node_idx, group_idx = _i32c(node_idx), _i32c(group_idx)
    w = {k: f(v) for k, v in w.items()}
    M, I, H = int(node_idx.shape[0]), int(x.shape[1]), int(w["wout"].shape[0])
    if int(group_idx.shape[1]) != 16 or not L.lib().roitr_local_td_supported(I, H, 16):
        raise L.RoitrError("local_td: in_dim 64, H 128, 16 neighbours per node")
    out = torch.empty((M, H), dtype=torch.float32, device=x.device)
    a = _LocalTd()
    a.M, a.in_dim, a.H = M, I, H
    a.x, a.node_idx, a.group_idx, a.ppf = L.ptr(x), L.ptr(node_idx), L.ptr(group_idx), L.ptr(ppf)
    no = f(node_order) if node_order is not None else None
    a.node_order = L.ptr(no)
    for k in ("wqqt", "bqqt", "wv", "bv", "wpe", "wvpe", "bvpe", "wcat", "bcat", "norm_w", "norm_b", "wout", "bout"):
        setattr(a, k, L.ptr(w[k]))
    a.scale, a.eps, a.out = 1.0 / float(H // 4) ** 0.5, 1e-5, L.ptr(out)
    L.check(L.lib().roitr_local_td(ctypes.byref(a), L.stream_ptr()), "local_td")
    return out


class _LocalBlock(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int), ("K", ctypes.c_int), ("H", ctypes.c_int),
                ("x", ctypes.c_void_p), ("kv", ctypes.c_void_p), ("group_idx", ctypes.c_void_p), ("ppf", ctypes.c_void_p),
                ("node_order", ctypes.c_void_p), ("wq", ctypes.c_void_p), ("bq", ctypes.c_void_p),
                ("wpe", ctypes.c_void_p), ("bpe", ctypes.c_void_p), ("wvpe", ctypes.c_void_p), ("bvpe", ctypes.c_void_p),
                ("wcat", ctypes.c_void_p), ("bcat", ctypes.c_void_p), ("norm_w", ctypes.c_void_p), ("norm_b", ctypes.c_void_p),
                ("wout", ctypes.c_void_p), ("bout", ctypes.c_void_p), ("bn2_w", ctypes.c_void_p), ("bn2_b", ctypes.c_void_p),
                ("scale", ctypes.c_float), ("eps", ctypes.c_float), ("out", ctypes.c_void_p), ("kv_bf16", ctypes.c_int),
                ("wq_h", ctypes.c_void_p), ("wcat_h", ctypes.c_void_p), ("wout_h", ctypes.c_void_p)]


def local_block(x, kv, group_idx, ppf, w, node_order=None, variant=None, bf16_weights=False):
    """The block form of the local PPF transformer in one launch (csrc/local_block.hip; model/model.py:131-142 around
    ppftransformer.py:227-253).  x (M, H), kv (M, 2H) = k | v rows of every point, group_idx (M, K) int32, ppf (M, K, 4);
    w: dict of the FOLDED weights (include/roitr_engine.h RoitrLocalBlock): wq (H,H) bq, wpe (H,4) bpe, wvpe (H,4) bvpe,
    wcat (H,2H) bcat, norm_w norm_b, wout (H,H) bout, bn2_w bn2_b.  node_order: optional (M, 4) float32 whose last column
    holds the node index bits (the grid's sorted-point array).  variant: tuning hook (1: no attention, 2: attention only).
    A bfloat16 `kv` tensor is gathered as stored (RoitrLocalBlock::kv_bf16, the engine's bf16 operand mode); bf16_weights (with it):
    the three on-chip GEMMs take bf16 matrix operands (RoitrLocalBlock::wq_h / wcat_h / wout_h; the copies are made here)."""
    M, H = int(x.shape[0]), int(x.shape[1])
    K = int(group_idx.shape[1])
    f = lambda t: t.contiguous().float()
    kv_h = kv.dtype == torch.bfloat16
    x, kv, ppf = f(x), (kv.contiguous() if kv_h else f(kv)), f(ppf)
    group_idx = _i32c(group_idx)
    keep = {k: f(v) for k, v in w.items()}
    out = torch.empty((M, H), dtype=torch.float32, device=x.device)
    a = _LocalBlock()
    a.M, a.K, a.H = M, K, H
    a.x, a.kv, a.group_idx, a.ppf = L.ptr(x), L.ptr(kv), L.ptr(group_idx), L.ptr(ppf)
    no = f(node_order) if node_order is not None else None
    a.node_order = L.ptr(no)
    for k in ("wq", "bq", "wpe", "bpe", "wvpe", "bvpe", "wcat", "bcat", "norm_w", "norm_b", "wout", "bout", "bn2_w", "bn2_b"):
        setattr(a, k, L.ptr(keep[k]))
    a.scale, a.eps, a.out = 1.0 / float(H // 4) ** 0.5, 1e-5, L.ptr(out)
    a.kv_bf16 = int(kv_h)
    if bf16_weights:
        keep_h = {k: keep[k].to(torch.bfloat16).contiguous() for k in ("wq", "wcat", "wout")}
        a.wq_h, a.wcat_h, a.wout_h = L.ptr(keep_h["wq"]), L.ptr(keep_h["wcat"]), L.ptr(keep_h["wout"])
    if variant is None:
        L.check(L.lib().roitr_local_block(ctypes.byref(a), L.stream_ptr()), "local_block")
    else:
        L.check(L.lib().roitr_local_block_dbg(ctypes.byref(a), int(variant), L.stream_ptr()), "local_block_dbg")
    return out
