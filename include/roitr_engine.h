/*
 * roitr_engine.h -- C ABI of libroitr_hip.so, part 2: the fused stage operators of the RoITr
 * inference path and the whole-pair engine (boundary B2 of SURVEY.md 8(b), restated as C entry
 * points so a host in any language can drive the path; the roitr_amd Python package is one).
 *
 * Conventions: device pointers to contiguous fp32 / int32; caller owns all memory; every call is
 * asynchronous on `stream` and returns 0 or an error code (text via roitr_last_error()).
 * Citations are into /root/reference.
 */
#ifndef ROITR_ENGINE_H
#define ROITR_ENGINE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef ROITR_POINTOPS_H
typedef struct ihipStream_t* roitr_stream_t;
#endif

/* ------------------------------------------------------------------ dense layers (nn.Linear) */
/* C[b] = act(alpha * (A[b] (+A2[b])) @ W[b]^T + bias[b]);  A (M,K) lda, W (N,K) ldw, C (M,N) ldc.
 * a_idx / w_idx: optional int32 row gathers; an index < 0 or >= *_limit (when *_limit > 0) reads
 * a zero row (the reference's zero-padded patch rows, model/RIGA_v2.py:129-142). */
typedef struct RoitrGemm {
    int M, N, K;
    const float* A; const float* A2; int lda; const int* a_idx; int a_limit;
    const float* W; int ldw; const int* w_idx; int w_limit;
    const float* bias; float alpha; int relu;
    float* C; int ldc;
    int batch; long sA, sW, sC, sBias, sAidx, sWidx;
    /* optional ragged batching: batch b multiplies row segment (seg_a0 + b) of A with row segment (seg_w0 + b) of W,
     * segments given by the cumulative int32 `seg_off`; M / N then only bound the grid. */
    const int* seg_off; int seg_a0, seg_w0;
    /* optional fused LayerNorm epilogue (ln_gamma != NULL; requires N == 64 = one tile per row, batch == 1):
     *   C[r,:] = [relu]( LayerNorm(acc[r,:] + bias + ln_res[ln_res_idx ? ln_res_idx[r] : r, :]) * gamma + beta + ln_post[r,:] )
     * i.e. the GEMM followed by roitr_add_layernorm in one launch (ln_res / ln_post rows are 64 floats, dense). */
    const float* ln_gamma; const float* ln_beta; const float* ln_res; const int* ln_res_idx; const float* ln_post;
    int ln_relu; float ln_eps;
    /* bf16 operand mode (0 = the fp32 kernel above): a mask of ROITR_BF16_*.  The products run on the bf16 matrix cores with
     * fp32 accumulation; W MUST be stored bf16 (ROITR_BF16_W: `W` then points to uint16 data, ldw / sW in elements), A is
     * fp32 and rounded while it is staged unless ROITR_BF16_A says it is stored bf16 (`A` -> uint16, A2 unsupported), C is
     * written bf16 (uint16, ldc / sC in elements) with ROITR_BF16_C.  bias / ln_* stay fp32.  Needs K % 64 == 0 and
     * 16-byte aligned rows (roitr_gemm_bf16_supported). */
    int bf16;
    /* optional K-concatenated A operand (fp32 kernel only): columns k >= k_cat of the product come from A_cat (same row
     * index / gather as A, leading dimension lda_cat), i.e. C = [A | A_cat] @ W^T with W (N, K), K = k_cat + width(A_cat);
     * k_cat % 32 == 0.  Lets `linear(att) + in_proj(x)` of a local transformer run as ONE GEMM. */
    const float* A_cat; int lda_cat; int k_cat;
    /* optional, fused LayerNorm epilogue of the fp32 kernel only: the three-nearest-neighbour interpolation of TransitionUp
     * (pointops.py:168-182 behind model/model.py:112-116) added AFTER the activation,
     *   C[r,:] += sum_k w_k ip_feat[ip_idx[3 r + k], :],   w_k = (1 / (sqrt(ip_dist2[3 r + k]) + 1e-8)) / sum of the three,
     * ip_feat rows N floats, dense: `linear1(x1) + interpolation(...)` in the launch that computes linear1. */
    const float* ip_feat; const int* ip_idx; const float* ip_dist2;
    /* optional with A_cat: its own row gather (A_cat row of output row r = a_cat_idx[r]; A keeps a_idx / the identity) -- the
     * `linear(att) + in_proj(x[node_idx])` of a TransitionDown transformer as one GEMM (model/model.py:59-62 samples the rows).
     * With A_cat an addend A2 applies to the A part only (columns k < k_cat). */
    const int* a_cat_idx;
    /* optional (ABI 3): device int; batches (tiles of batch index) >= *batch_live leave at once -- a batch list whose live length is
     * only known on the device (the compacted patch list of the adaptive matching, RIGA_v2.py:126-152) without a host round trip. */
    const int* batch_live;
    /* ROITR_BF16_X3 (ABI 3; alone in `bf16`): fp32 arithmetic on the bf16 matrix cores by a three-way operand split (csrc/gemm_x3.hip).
     * `W` points to THREE bf16 images of the fp32 weight (roitr_split_bf16x3): piece p of element (n, k) at W[p * w_piece + n * ldw + k]
     * (uint16, ldw / w_piece in elements); A (A2, A_cat) stay fp32 and are split while they are staged; C fp32.  Six bf16 products per
     * multiply (error ~ one fp32 rounding per product; rows bitwise independent of the row count).  Needs K % 32 == 0, batch == 1. */
    long w_piece;
} RoitrGemm;
#define ROITR_BF16_W 1
#define ROITR_BF16_A 2
#define ROITR_BF16_C 4
#define ROITR_BF16_X3 8
int roitr_gemm(const RoitrGemm* g, roitr_stream_t stream);
int roitr_gemm_bf16_supported(const RoitrGemm* g);
/* fp32 -> bf16 (round to nearest even), n elements; dst 4-byte aligned */
int roitr_f32_to_bf16(long n, const float* src, unsigned short* dst, roitr_stream_t stream);
/* shapes / layouts the split kernel takes (RoitrGemm::bf16 == ROITR_BF16_X3) */
int roitr_gemm_x3_supported(const RoitrGemm* g);
/* src (n fp32) -> its three bf16 pieces at dst, dst + piece, dst + 2 * piece (piece >= n, even): src[i] == h[i] + m[i] + l[i] exactly */
int roitr_split_bf16x3(long n, const float* src, unsigned short* dst, long piece, roitr_stream_t stream);

/* ------------------------------------------------------------------ row-wise layers */
/* out = act( LayerNorm(x + res[res_idx]) * gamma + beta (+ post_add) ); res, res_idx, post_add optional.
 * attention.py:319, model/model.py:138-140 (bn2, += identity, relu), geoattention.py:50,161,241. */
int roitr_add_layernorm(int M, int C, const float* x, const float* res, const int* res_idx, const float* gamma,
                        const float* beta, const float* post_add, int relu, float eps, float* out, roitr_stream_t stream);
/* the same followed by the three-nearest-neighbour interpolation of TransitionUp, added after the activation (see RoitrGemm::ip_*):
 * out[r,:] = act( LayerNorm(x[r,:] + res) * gamma + beta ) + sum_k w_k ip_feat[ip_idx[3 r + k], :] */
int roitr_add_layernorm_interp(int M, int C, const float* x, const float* res, const int* res_idx, const float* gamma,
                               const float* beta, int relu, float eps, const float* ip_feat, const int* ip_idx, const float* ip_dist2,
                               float* out, roitr_stream_t stream);
/* F.normalize(p=2, dim=1), model/RIGA_v2.py:64-65 */
int roitr_l2_normalize(int M, int C, const float* x, float* out, roitr_stream_t stream);
int roitr_transpose(int rows, int cols, const float* in, int ld_in, float* out, int ld_out, roitr_stream_t stream);
/* out = a + b over n floats (bias of a folded layer) */
int roitr_add_vectors(int n, const float* a, const float* b, float* out, roitr_stream_t stream);
/* out = base + 3-NN inverse-distance interpolation of feat (functions/pointops.py:168-182 + model/model.py:116);
 * dist2 = SQUARED distances as produced by the kNN call. */
int roitr_interp3_add(int n, int C, const float* feat, const int* idx, const float* dist2, const float* base, float* out,
                      roitr_stream_t stream);
/* per-cloud feature mean (model/model.py:101-109) */
int roitr_segment_mean(int b, int C, const float* x, const int* offset, float* out, roitr_stream_t stream);
/* SinusoidalPositionalEmbedding (positional_encoding.py:38-62): out (rows, C) */
int roitr_sinusoid(long rows, int C, const float* vals, const float* div_term, float* out, roitr_stream_t stream);
/* Fused positional_encoding.py:139-154: out[r,:] = proj_d(sinusoid(d_idx[r])) + max_k proj_a(sinusoid(a_idx[r,k])) */
int roitr_geo_embed(long rows, int C, int angle_k, const float* d_idx, const float* a_idx, const float* div_term,
                    const float* Wd, const float* bd, const float* Wa, const float* ba, float* out, roitr_stream_t stream);
/* OPT-IN: the same embedding on the bf16 matrix cores with three-way split operands (x = hi + mid + lo, six partial
 * products kept: error of the order of fp32 rounding).  W*3: 3 * C * C uint16 made by roitr_split3_bf16. */
int roitr_split3_bf16(long n, const float* src, unsigned short* dst, roitr_stream_t stream);
int roitr_geo_embed_split(long rows, int C, int angle_k, const float* d_idx, const float* a_idx, const float* div_term,
                          const unsigned short* Wd3, const float* bd, const unsigned short* Wa3, const float* ba, float* out,
                          roitr_stream_t stream);
/* bf16 operand form (engine operand_dtype = 1): Wd / Wa = the (C, C) weights stored bf16, sinusoid rounded to bf16, fp32 accumulate */
int roitr_geo_embed_bf16(long rows, int C, int angle_k, const float* d_idx, const float* a_idx, const float* div_term,
                         const unsigned short* Wd, const float* bd, const unsigned short* Wa, const float* ba, float* out,
                         roitr_stream_t stream);
/* the same with the embedding itself STORED in bf16 (out: rows x C uint16) -- half the bytes of the tensor every self
 * layer of the global transformer streams (read back by roitr_mha with e_bf16 = 1) */
int roitr_geo_embed_bf16_out(long rows, int C, int angle_k, const float* d_idx, const float* a_idx, const float* div_term,
                             const unsigned short* Wd, const float* bd, const unsigned short* Wa, const float* ba,
                             unsigned short* out, roitr_stream_t stream);
/* Function-table form of the same embedding (csrc/geo_table.hip): proj_x(sinusoid(v)) is a univariate band-limited function of
 * the scalar v per output channel (positional_encoding.py:38-62 feeds ONE value per row into the sinusoid), so it is fitted once
 * per weight set by a degree-7 polynomial per channel on intervals of width `interval` (float64 Chebyshev interpolation on the
 * HOST) and evaluated from LDS.  roitr_geo_table_build: all pointers are HOST memory; table holds roitr_geo_table_floats()
 * floats; fit[0..5] = {max |poly - g_d| over a dense probe (64 points per interval), max |g_d|, the same for the angle projection,
 * the largest per-channel relative error of the distance / of the angle projection}.
 * roitr_geo_embed_table: device pointers; values outside [0, n_int * interval) are evaluated directly from div_term / W / b, so
 * any input is served; angle_k must be 3, C a multiple of 64; out is fp32 (rows, C), or bf16 when out_bf16. */
size_t roitr_geo_table_floats(int C, int n_int_d, int n_int_a);
int roitr_geo_table_build(int C, const float* div_term, const float* Wd, const float* bd, const float* Wa, const float* ba,
                          float interval, int n_int_d, int n_int_a, float* table, double* fit);
int roitr_geo_embed_table(long rows, int C, int angle_k, const float* d_idx, const float* a_idx, const float* table, float interval,
                          int n_int_d, int n_int_a, const float* div_term, const float* Wd, const float* bd, const float* Wa,
                          const float* ba, void* out, int out_bf16, roitr_stream_t stream);
/* E = P_d + max_k P_a[:, k, :] (positional_encoding.py:146-152) */
int roitr_geo_combine(long rows, int C, int k, const float* pd, const float* pa, float* out, roitr_stream_t stream);
int roitr_gather_rows(long rows, int C, const float* in, const int* idx, int limit, float* out, roitr_stream_t stream);
int roitr_compose_idx(int n, const int* outer, const int* inner, int* out, roitr_stream_t stream);
/* lib/utils.py:358-389 with the gather fused: out (m,k,4) */
int roitr_calc_ppf(int m, int k, const float* centre_xyz, const float* centre_normals, const float* ref_xyz,
                   const float* ref_normals, const int* group_idx, float* out, roitr_stream_t stream);

/* ------------------------------------------------------------------ local PPF attention */
/* attention.py:152-200 with the positional branch folded (see csrc/local_attn.hip).
 * q: (M, >= H + 5*heads) rows = [q | per head: Wpe_h^T q_h (4), q_h.bpe_h (1)] (or (M, >= H) rows with wpe / bpe given); k, v: rows of the input
 * cloud (ld given); group_idx (M,K) int32; ppf (M,K,4); wvpe (H,4), bvpe (H); out (M,H).
 * scale = 1/sqrt(H/heads). */
typedef struct RoitrLocalAttn {
    int M, K, H, heads;
    const float* q; int ldq;
    const float* k; int ldk;
    const float* v; int ldv;
    const int* group_idx; const float* ppf;
    const float* wvpe; const float* bvpe;
    float scale;
    float* out; int ldo;
    const void* node_order;   /* optional float4[M] (x,y,z,index-as-bits): visiting order, e.g. roitr_knn_sorted_points() */
    int bf16;                 /* 0: fp32 rows; 3: q / k / v rows AND the output row are stored in bf16 (uint16; ld* in elements) --
                                 the engine's bf16 operand mode, where the q|k|v tensor is written bf16 by its GEMM */
    const float* wpe; const float* bpe;   /* optional (both or none): Wpe (H,4) and bpe (H) of the folded positional branch; the
                                 kernel then forms qp[h] = [Wpe_h^T q_h, q_h . bpe_h] itself and q rows are only H wide */
} RoitrLocalAttn;
int roitr_local_attention(const RoitrLocalAttn* a, roitr_stream_t stream);
int roitr_build_pfold(int H, int heads, const float* wpe, const float* bpe, float* pfold, roitr_stream_t stream);

/* TransitionDown form of the local PPF attention with the key / value projections folded into the query side
 * (csrc/local_attn.hip local_attn_fold_kernel; attention.py:152-200 behind ppftransformer.py:227-253).  M query nodes, 16 neighbours
 * each among the rows of x (N_in, in_dim):
 *     score(h, j) = scale * ( qt[node][h] . x_j + (Wpe_h^T q_h) . ppf_j )       (terms constant over j are dropped: softmax)
 *     xbar[node][h] = sum_j a_hj x_j            (M, 4, in_dim)  -> the caller applies Wv'_h and bv'_h (one batched GEMM)
 *     vpart[node]   = Wvpe_h pbar_h + bvpe_h    (M, H)          -> the positional value term; attention output = vpart + Wv' xbar + bv'
 * with qt[node][h] = Wk'_h^T q_h (M, 4, in_dim) supplied by the caller.  fp32, 4 heads, K = 16,
 * (in_dim, H) in {(64, 128), (128, 256), (256, 256)}; everything 16-byte aligned. */
typedef struct RoitrLocalAttnFold {
    int M, in_dim, H;
    const float* x; int ldx;
    const float* q; int ldq;      /* (M, H) query rows (for the PPF coefficients) */
    const float* qt;              /* (M, 4 * in_dim) */
    const int* group_idx; const float* ppf;   /* (M, 16), (M, 16, 4) */
    const float* wpe;             /* (H, 4) */
    const float* wvpe; const float* bvpe;     /* (H, 4), (H) */
    float scale;
    float* xbar;                  /* (M, 4 * in_dim) */
    float* vpart;                 /* (M, H) */
    const void* node_order;       /* optional float4[M]: visiting order */
    int ldqt;                     /* row stride of qt in floats; 0 = dense (4 * in_dim).  Lets q and q~ be the two column blocks of ONE
                                   * GEMM's output (engine, round 5: q~_h = (Wk'_h^T Wq'_h) x + Wk'_h^T bq_h straight from the input row) */
} RoitrLocalAttnFold;
int roitr_local_attention_fold(const RoitrLocalAttnFold* a, roitr_stream_t stream);
int roitr_local_attention_fold_supported(int in_dim, int H, int K);

/* The block form of the local PPF transformer in ONE launch (csrc/local_block.hip), for the 64- / 128-wide levels:
 *   out = relu( bn2( out_proj( LN( linear(att) + in_proj(x) ) ) ) + x )     model/model.py:131-142, ppftransformer.py:227-253
 * with att = local PPF attention of q = x Wq^T + bq over the neighbours' k | v rows.  The caller supplies kv (M, 2H) = the k | v
 * projections of EVERY point (one plain GEMM) and the folded weights: wq / bq = proj_q o in_proj, wcat (H, 2H) = [W_linear | W_in],
 * bcat = b_linear + b_in, wpe / bpe / wvpe / bvpe = the folded positional branch (see RoitrLocalAttn).  fp32, heads = 4,
 * H in {64, 128}, K in {8, 16}; everything 16-byte aligned. */
typedef struct RoitrLocalBlock {
    int M, K, H;
    const float* x; const float* kv; const int* group_idx; const float* ppf;
    const void* node_order;   /* optional float4[M] (x,y,z,index-as-bits): visiting order */
    const float* wq; const float* bq;
    const float* wpe; const float* bpe; const float* wvpe; const float* bvpe;
    const float* wcat; const float* bcat; const float* norm_w; const float* norm_b;
    const float* wout; const float* bout; const float* bn2_w; const float* bn2_b;
    float scale, eps;
    float* out;
    int kv_bf16;   /* ABI 3: 1 = the k | v rows are STORED in bf16 (`kv` -> uint16, 2 H elements per row); everything else stays fp32 */
    /* ABI 4 (optional, all three or none, with kv_bf16 = 1): bf16 copies of wq (H x H), wcat (H x 2H), wout (H x H).  Given: the three
     * on-chip GEMMs take bf16 matrix operands (v_mfma_f32_32x32x16_bf16, fp32 accumulate; the activations are rounded on their way into
     * the operand registers, like the A operand of a ROITR_BF16_W GEMM) -- the fused block of the engine's bf16 operand mode */
    const unsigned short* wq_h; const unsigned short* wcat_h; const unsigned short* wout_h;
} RoitrLocalBlock;
int roitr_local_block(const RoitrLocalBlock* a, roitr_stream_t stream);
int roitr_local_block_supported(int H, int K);

/* The TransitionDown transformer of the 64 -> 128 wide level in ONE launch (csrc/local_block.hip local_td_kernel, round 5):
 *   out = out_proj( LN( linear(att) + in_proj(x_n) ) ),  x_n = x[node_idx[node]],                      ppftransformer.py:227-253
 * att = the folded-attention form of RoitrLocalAttnFold (scores from q~_h = Wk'_h^T q_h against the 16 gathered INPUT rows, value
 * = Wv'_h (sum_j a_hj x_j) + bv'_h + the positional value term).  A workgroup keeps a tile of 32 nodes on chip from x_n to out: q | q~
 * (one on-chip GEMM with the folded weight wqqt), the attention, the per-head value projection, the K-concatenated linear with its
 * LayerNorm, out_proj.  fp32, 4 heads, in_dim = 64, H = 128, K = 16; weights as the engine folds them (csrc/engine.cpp fold_local):
 * wqqt ((H + 4 in_dim), in_dim) / bqqt, wv (H, in_dim) / bv = the folded value projection, wcat (H, H + in_dim) = [W_linear | W_in] /
 * bcat, wpe / wvpe (H, 4) / bvpe the folded positional branch.  Everything 16-byte aligned. */
typedef struct RoitrLocalTd {
    int M, in_dim, H;
    const float* x; const int* node_idx;      /* (N_in, in_dim) input rows; (M) row of x of every node */
    const int* group_idx; const float* ppf;   /* (M, 16) rows of x, (M, 16, 4) */
    const void* node_order;                   /* optional float4[M]: visiting order */
    const float* wqqt; const float* bqqt;
    const float* wv; const float* bv;
    const float* wpe; const float* wvpe; const float* bvpe;
    const float* wcat; const float* bcat; const float* norm_w; const float* norm_b;
    const float* wout; const float* bout;
    float scale, eps;
    float* out;                               /* (M, H) */
} RoitrLocalTd;
int roitr_local_td(const RoitrLocalTd* a, roitr_stream_t stream);
int roitr_local_td_supported(int in_dim, int H, int K);

/* The first local transformer of the network (in_planes = 1, model/model.py:152): its q | k | v are rank-1 affine in the scalar
 * input feature, so the whole TransitionDown transformer collapses to per-node scalars + two small on-chip GEMMs
 * (csrc/local_block.hip local_first_kernel).  x (M,) the scalar feature; H = 64.  Constants (built from the layer's weights, see
 * csrc/engine.cpp build_local_first): head_consts (4 heads x 16 floats: c1 c2 c3 c4 | P1[4] | P0[4] | d1 d0 | 0 0),
 * G (64, 32) = the affine map of g = [S(4) | pbar(16) | x | 1 | 0...] to linear(att) + in_proj(x), zero_bias (64 zeros). */
typedef struct RoitrLocalFirst {
    int M, K;
    const float* x; const int* group_idx; const float* ppf; const void* node_order;
    const float* head_consts; const float* G; const float* zero_bias;
    const float* norm_w; const float* norm_b; const float* wout; const float* bout;
    float scale, eps;
    float* out;
} RoitrLocalFirst;
int roitr_local_first(const RoitrLocalFirst* a, roitr_stream_t stream);

/* ------------------------------------------------------------------ global geometric transformer */
/* positional_encoding.py:110-137 get_embedding_indices for a batch of clouds.  pts (rows,3) = all nodes,
 * offset (b) cumulative, cloud_of_row (rows), eoff (b) = element offset of cloud c's (n_c, n_c) block.
 * d_idx: concatenated (n_c, n_c); a_idx: concatenated (n_c, n_c, angle_k). */
int roitr_geo_indices(int rows, const float* pts, const int* offset, const int* cloud_of_row, const long* eoff,
                      float sigma_d, float sigma_a, int angle_k, int n_max, float* d_idx, float* a_idx, roitr_stream_t stream);

/* Multi-head attention, query rows [q_row0, q_row0 + q_rows) of the concatenated node array.
 * Keys/values of query row r are the rows of cloud partner[cloud_of_row[r]] (partner == NULL: own cloud).
 * Without E: geoattention.py:26-66 (cross attention; q/k already hold the +pos inputs).
 * With E (n_c,n_c,C per cloud at eoff): geoattention.py:87-136 with the RPE branch folded --
 *   qt (rows, heads, C) = Wp_h^T q_h,  bp (C) = proj_p.bias,  ebar (rows, heads, C) = sum_j a'_ij E_ij
 *   where a' is the diagonal-masked softmax; the caller finishes pos_states = Wvp_h ebar_h + bvp_h. */
typedef struct RoitrMha {
    int q_row0, q_rows, C, heads;
    const float* q; int ldq;
    const float* k; int ldk;
    const float* v; int ldv;
    const int* offset; const int* cloud_of_row; const int* partner;
    const float* E; const long* eoff; const float* qt; const float* bp;
    float scale; int nk_max;
    float* out; int ldo;
    float* ebar;
    int e_bf16;   /* E is stored in bf16 (uint16, same element offsets): engine operand_dtype = bf16; needs C = 256 or 512, 4 heads */
} RoitrMha;
int roitr_mha(const RoitrMha* a, roitr_stream_t stream);

/* ------------------------------------------------------------------ coarse-to-fine matching tail */
/* Cloud layout for every batched call below: [src_0 .. src_{B-1}, tgt_0 .. tgt_{B-1}], B = pairs. */

/* lib/utils.py:428-471.  p2n (n_points) node index local to the cloud; knn_idx (n_nodes, limit) point index
 * local to the cloud, padded with the cloud's point count; masks are int32 0/1. */
int roitr_point_to_node_partition(int b, int n_points, int n_nodes, const float* pts, const int* pt_offset,
                                  const float* nodes, const int* node_offset, const int* cloud_of_node, int limit,
                                  int* p2n, float* p2n_dist, int* node_masks, int* knn_idx, int* knn_mask, roitr_stream_t stream);

/* model/modules.py:141-178 CoarseMatching (ref = tgt, src = src as called at RIGA_v2.py:121).
 * feats (n_nodes, C) L2-normalised; outputs (pairs, num_corr): node indices local to their cloud (-1 beyond
 * n_corr[pair]), scores.  scratch: pairs * scratch_stride floats, stride >= roitr_coarse_scratch_floats().
 * xy: optional precomputed feature dot products, pair b at xy + b*xy_stride, row-major (n_tgt, ld = xy_ld). */
typedef struct RoitrCoarse {
    int pairs, C, num_corr, dual_norm, max_ref, max_src;
    const float* feats; const int* node_offset; const int* node_masks;
    float* scratch; long scratch_stride;
    int* tgt_corr; int* src_corr; float* corr_scores; int* n_corr;
    const float* xy; long xy_stride; int xy_ld;
    int lds_cap;   /* filled in by the launcher (LDS keys per sort chunk); callers leave it 0 */
} RoitrCoarse;
size_t roitr_coarse_scratch_floats(int n_ref, int n_src);
int roitr_coarse_matching(const RoitrCoarse* a, roitr_stream_t stream);
/* model/modules.py:75-124 AdaptiveSuperPointMatching (4DMatch); needs a->xy; a->num_corr = output capacity per pair */
int roitr_adaptive_matching(const RoitrCoarse* a, int min_num, float threshold, roitr_stream_t stream);

/* Patch layouts of the four operators below (ABI 3).  STRIDED (pair_off == NULL): per-patch arrays have pairs * num_corr slots,
 * patch p of pair b at slot b * num_corr + p, live while p < n_corr[b] (dead slots are zero-filled / skipped).  COMPACTED
 * (pair_off != NULL, pairs + 1 cumulative int32 from roitr_patch_offsets): the live patches of all pairs back to back in `slots`
 * slots, pair b owns [pair_off[b], pair_off[b + 1]); slots past pair_off[pairs] are never read or written.  That is how the
 * reference runs the tail -- on the SELECTED node pairs only (model/RIGA_v2.py:126-152, modules.py:102-111): the adaptive 4DMatch
 * matching selects between 128 and n_t * n_s = 15 625 pairs per cloud pair, typically ~1 000.  The coarse lists tgt_corr /
 * src_corr / corr_scores stay (pairs, num_corr) in both layouts. */
/* pair_off[0] = 0, pair_off[b + 1] = min(slots, n_corr[0] + .. + n_corr[b]); the caller detects a cut from n_corr */
int roitr_patch_offsets(int pairs, const int* n_corr, int slots, int* pair_off, roitr_stream_t stream);

/* model/RIGA_v2.py:125-147: per patch correspondence the `limit` point rows / points / masks of both sides.
 * rows index the concatenated point arrays (-1 = the zero pad row). */
typedef struct RoitrPatch {
    int pairs, num_corr, limit;
    const int* n_corr; const int* tgt_corr; const int* src_corr;
    const int* node_offset; const int* pt_offset; const int* knn_idx; const int* knn_mask; const float* points;
    int* tgt_rows; int* src_rows; int* tgt_masks; int* src_masks; float* tgt_pts; float* src_pts;
    const int* pair_off; int slots;   /* compacted layout (see above); NULL / 0 = strided */
} RoitrPatch;
int roitr_patch_gather(const RoitrPatch* a, roitr_stream_t stream);

/* model/modules.py:10-72 LearnableLogOptimalTransport: scores (patches, limit, limit) -> out (patches, limit+1, limit+1) */
typedef struct RoitrOT {
    int pairs, num_corr, limit, num_iter;
    const int* n_corr; const float* scores; const int* row_masks; const int* col_masks; const float* alpha;
    float* out;
    const int* pair_off; int slots;   /* compacted layout; NULL / 0 = strided */
} RoitrOT;
int roitr_optimal_transport(const RoitrOT* a, roitr_stream_t stream);
/* Diagnostics of the data-dependent work of that stage (synchronous): reads out[0] = live patches, out[1] = Sinkhorn iterations skipped by
 * the exact fixed-point exit, out[2] = patches served by the log-domain kernel, all since the last enable; then enable = 1 (re)starts
 * counting from zero, 0 stops it, -1 leaves it.  out may be NULL. */
int roitr_ot_stats(int enable, unsigned long long* out);

/* model/modules.py:216-324 FineMatching (use_dustbin = False).  ot = the OT output; rows = tgt, cols = src.
 * Emits correspondences in (patch, row, col) order; offsets[patch] = first output slot of the patch. */
typedef struct RoitrFine {
    int pairs, num_corr, limit, k, mutual;
    float conf;
    const int* n_corr; const float* ot; const int* row_masks; const int* col_masks;
    const float* row_pts; const float* col_pts; const float* global_scores;
    unsigned char* flags; int* counts; int* offsets; int* n_out;
    float* out_row_pts; float* out_col_pts; float* out_scores; int* out_patch;
    long out_cap;   /* rows the out_* arrays hold (0 = caller guarantees pairs*num_corr*limit*limit): the emitter never writes past
                       it and *n_out is clamped to it.  Worst case per patch: limit*k rows when mutual, 2*limit*k otherwise
                       (row top-k OR column top-k, modules.py:259-266). */
    const int* pair_off; int slots;   /* compacted layout; NULL / 0 = strided.  out_patch then holds slot numbers */
    int* pair_starts;                 /* optional (pairs + 1): first output row of every pair, then the total */
} RoitrFine;
int roitr_fine_matching(const RoitrFine* a, roitr_stream_t stream);

/* ------------------------------------------------------------------ ground-truth side outputs (need rot/trans) */
/* Padded clouds for lib/utils.py:506-510: out_pts has n_points + 2*pairs rows -- every cloud followed by its pad row
 * (the zero row of RIGA_v2.py:86-87), source clouds (and their pad rows) transformed by rot/trans;
 * out_offset (3*pairs): cumulative padded sizes of the 2*pairs clouds, then the pairs target offsets relative to
 * the first target row. */
int roitr_build_padded_clouds(int pairs, int n_points, const float* pts, const int* pt_offset, const float* rot,
                              const float* trans, float* out_pts, int* out_offset, roitr_stream_t stream);
/* lib/utils.py:511-527.  d2_padded: squared distance of every padded point to its nearest neighbour in the
 * partner cloud (kNN(1) over the padded clouds). */
int roitr_node_occlusion_score(int n_nodes, int limit, const int* cloud_of_node, const int* pt_offset, const int* knn_idx,
                               const int* knn_mask, const int* node_masks, const float* d2_padded, float overlap_thres,
                               float* out, roitr_stream_t stream);
/* lib/utils.py:530-614 (ref = tgt, src = src).  overlap: scratch (pairs, mat_stride >= max_nodes^2);
 * out_idx (pairs, mat_stride, 2) [ref, src] local node indices in torch.nonzero order, out_overlap, out_count (pairs).
 * During the call out_idx / out_count[0] also carry the work list of node pairs that survive the enclosing-sphere prune (they are
 * rewritten by the final compaction); pairs * max_nodes^2 must stay below 2^31. */
typedef struct RoitrNodeCorr {
    int pairs, limit, max_nodes;
    float pos_radius;
    const float* nodes; const int* node_offset; const int* node_masks;
    const float* points; const int* pt_offset; const int* knn_idx; const int* knn_mask;
    const float* rot; const float* trans;
    float* overlap; long mat_stride;
    int* out_idx; float* out_overlap; int* out_count;
    int n_nodes; float* nodes_t; float* radius;   /* scratch: (n_nodes,3) and (n_nodes) */
} RoitrNodeCorr;
int roitr_node_correspondences(const RoitrNodeCorr* a, roitr_stream_t stream);

/* ------------------------------------------------------------------ the whole-pair engine */
/* One engine = one set of weights + its workspace on the current device.  Drives the complete test-mode
 * forward of model/RIGA_v2.py:58-175 for a batch of B independent pairs with HIP kernels only.
 *
 * Parameters are registered by their reference state_dict key (e.g.
 * "backbone.enc2.0.transformer.transformer.attention.proj_p.weight", lib/trainer.py:94-130 layout) as
 * device pointers that must stay valid while the engine lives. */
typedef struct RoitrEngineConfig {
    int factor;            /* 1 = 3DMatch, 2 = 4DMatch (model/RIGA_v2.py:24,28) */
    int num_corr;          /* num_est_coarse_corr */
    int point_limit;       /* point_per_patch (64) */
    int fine_topk, fine_mutual, fine_use_global_score;
    float fine_conf;       /* fine_matching_confidence_threshold */
    int n_geo_layers;      /* len(transformer_architecture) */
    int geo_is_cross[16];  /* 0 = 'self', 1 = 'cross' */
    float matching_radius; /* coarse_matching.matching_radius (GT helpers) */
    int adaptive_coarse;   /* 1 = AdaptiveSuperPointMatching (4DMatch): num_corr is then its min_num_correspondences and
                              the per-pair patch capacity becomes n_tgt_nodes_max * n_src_nodes_max */
    float occlusion_radius;/* lib/utils.py:485 overlap_thres */
    int operand_dtype;     /* 0 = fp32 everywhere (the reference's arithmetic); 1 = bf16 operand storage for the dense layers
                              (BASELINE config 4): weights stored bf16 once at finalize, activations rounded to bf16 as MFMA
                              operands, GEMM-to-GEMM intermediates, the q|k|v tensors, the attention outputs and the geometric
                              embedding E stored bf16, the patch score contraction (RIGA_v2.py:150) on a bf16 copy of the point
                              descriptors; fp32 accumulation / bias / LayerNorm / softmax everywhere; FPS, kNN, PPF, partition,
                              coarse scores, optimal transport and fine matching stay fp32;
                              2 = fp32 everywhere like 0, but the plain linear layers with K >= 256 (levels 3 - 4, the global transformer,
                              the decoder's coarse half) multiply on the bf16 matrix cores by the three-way operand split of
                              csrc/gemm_x3.hip: fp32-accurate products (error against float64 not above the fp32 MFMA kernel's), rows
                              bitwise independent of the batch; results differ from mode 0 in the last bits only (round 6) */
} RoitrEngineConfig;

typedef struct RoitrForwardIO {
    int pairs;                 /* B */
    const int* n_points;       /* host array, 2B entries: src_0..src_{B-1}, tgt_0..tgt_{B-1} */
    /* inputs, device, clouds concatenated in that order */
    const float* points_geom;  /* (T,3): src_raw_pcd / tgt_pcd (model/RIGA_v2.py:62) */
    const float* normals;      /* (T,3) */
    const float* feats;        /* (T,1) */
    const float* points_out;   /* (T,3): src_pcd / tgt_pcd; NULL = points_geom */
    const float* rot;          /* (B,3,3) or NULL: skips the ground-truth side outputs */
    const float* trans;        /* (B,3) */
    /* outputs, device, caller-allocated; any may be NULL.  T4 = total nodes, P = num_corr, L = point_limit.  The per-patch outputs
     * -- *_knn_pts, *_knn_masks, matching_scores, fine_offsets -- have B * P slots, patch p of pair b at slot b * P + p, EXCEPT with
     * adaptive_coarse (4DMatch): there P = nmax4^2 only bounds the coarse lists tgt_corr / src_corr / corr_scores, and the per-patch
     * outputs hold the SELECTED patches of all pairs back to back in `patch_slots` slots (pair b at patch_offsets[b] ..
     * patch_offsets[b + 1]); see patch_slots below. */
    float* node_xyz;           /* (T4,3) */
    float* node_feats;         /* (T4, 256f) */
    float* point_feats;        /* (T, 256f) */
    int* node_masks;           /* (T4) */
    int* node_knn_idx;         /* (T4, L) local point indices, pad = cloud size */
    int* node_knn_mask;        /* (T4, L) */
    int* tgt_corr; int* src_corr; float* corr_scores; int* n_corr;   /* (B,P) x3, (B) */
    float* tgt_knn_pts; float* src_knn_pts;    /* (B,P,L,3) */
    int* tgt_knn_masks; int* src_knn_masks;    /* (B,P,L) */
    float* matching_scores;    /* (B,P,L+1,L+1) */
    float* out_tgt_pts; float* out_src_pts; float* out_scores; int* out_patch;  /* capacity B*P*L*fine_topk rows (x2 when fine_mutual == 0) */
    int* fine_offsets;         /* (B*P) first output row of every patch */
    int* n_out;                /* (1) total correspondences */
    float* gt_node_occ;        /* (T4) node occlusion scores, src clouds then tgt clouds; needs rot/trans */
    int* gt_corr_idx;          /* (B, nmax4^2, 2) [tgt node, src node]; needs rot/trans */
    float* gt_corr_overlaps;   /* (B, nmax4^2) */
    int* gt_corr_count;        /* (B) */
    /* Optional hipEvent_t.  NULL: the inputs above are ordered on `stream` (whatever produced them was queued there before the call).
     * Non-NULL: the inputs are complete once this event has fired and `stream` carries no dependency on them -- the engine then
     * stages the descriptors and runs the first sampling level (the longest serial chain of a small-batch forward) on its geometry
     * stream as soon as the event fires, BESIDE the previous forward still occupying `stream`, instead of behind it (calls of up to 128
     * pairs: the whole geometry chain, in scratch the engine alternates between calls); everything that touches the engine's shared
     * scratch -- and every kernel that writes one of the OUTPUT buffers above -- still starts where this forward begins on `stream`:
     * output buffers may be reused from call to call, nothing but `stream` order is required of the caller.  Results are identical.
     * Ignored by roitr_engine_forward_graph. */
    void* inputs_ready;
    /* ABI 3.  patch_slots: with adaptive_coarse, the number of patch slots the per-patch outputs above (and the engine's own patch
     * scratch) hold for the WHOLE call; 0 = the bound B * nmax4^2.  Patches beyond it are cut from the END of the call's list: the
     * true counts are in n_corr, the kept ones in patch_offsets -- a caller that sees sum(n_corr) > patch_slots repeats the call with
     * that sum (roitr_amd/riga.py does).  Ignored without adaptive_coarse (the slots are exactly B * num_corr there).
     * patch_offsets (B + 1): first patch slot of every pair, then the live total.  pair_starts (B + 1): first row of every pair in
     * out_scores / out_*_pts, then the total (== *n_out).  Both optional. */
    int* patch_offsets;
    int* pair_starts;
    int patch_slots;
} RoitrForwardIO;

void* roitr_engine_create(const RoitrEngineConfig* cfg);
void roitr_engine_destroy(void* engine);
int roitr_engine_set_param(void* engine, const char* name, const float* device_ptr, long numel);
int roitr_engine_finalize(void* engine, roitr_stream_t stream);
/* The function table of the geometric embedding chosen by finalize (see roitr_geo_table_build): info[0..8] = {interval, n_int_d,
 * n_int_a, fit error d, amplitude d, fit error a, amplitude a, largest per-channel relative error d, a}; returns 1 when a table is
 * in use, 0 when the GEMM form is.  finalize accepts the widest interval of {2, 1, 0.5} whose per-channel relative errors are both
 * below 2^-25, the GEMM form (geo_embed_kernel) otherwise. */
int roitr_engine_geo_table_info(void* engine, double* info);
int roitr_engine_forward(void* engine, const RoitrForwardIO* io, roitr_stream_t stream);
/* Same forward, replayed as ONE hipGraphLaunch once the same (sizes, io buffers) combination has been seen twice
 * (1st call: plain forward; 2nd: capture + instantiate; then replay).  `stream` must be a real (non-NULL) stream.
 * The caller keeps the io buffers at fixed addresses and re-fills them between calls. */
int roitr_engine_forward_graph(void* engine, const RoitrForwardIO* io, roitr_stream_t stream);
int roitr_engine_graph_count(void* engine);
/* Test taps: after stage `name` its output tensor is copied to `device_ptr` (NULL removes the tap);
 * inject: the stage output is REPLACED by the tensor at device_ptr before the forward continues. */
int roitr_engine_set_tap(void* engine, const char* name, void* device_ptr);
int roitr_engine_set_inject(void* engine, const char* name, const void* device_ptr);
/* level sizes the engine will use for a cloud of n points: out[0..3] (model/model.py:59-62 floor rule) */
void roitr_level_sizes(int n, int* out4);


/* ------------------------------------------------------------------ evaluators, batched over pairs
 * lib/loss.py:195-206 Evaluator.evaluate_fine == registration/benchmark_utils.py:69-77: pair b owns correspondences
 * [starts[b], starts[b+1]) of src_pts / tgt_pts; inliers[b] = #{ ||rot_b s + trans_b - t|| < radius }. */
int roitr_inlier_counts(int pairs, const int* starts, const float* src_pts, const float* tgt_pts, const float* rot,
                        const float* trans, float radius, int* inliers, roitr_stream_t stream);
/* lib/loss.py:170-193 Evaluator.evaluate_coarse: hits[b] = #{ i < n_corr[b] : (tgt_corr[b,i], src_corr[b,i]) is one of
 * the gt_count[b] ground-truth node pairs gt_idx[b,:,(tgt,src)] whose overlap exceeds acceptance_overlap }. */
int roitr_coarse_hits(int pairs, int num_corr, const int* n_corr, const int* tgt_corr, const int* src_corr, int gt_cap,
                      const int* gt_idx, const float* gt_overlaps, const int* gt_count, float acceptance_overlap, int* hits,
                      roitr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ROITR_ENGINE_H */
