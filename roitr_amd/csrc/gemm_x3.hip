// fp32 linear layers on the bf16 matrix cores by a three-way operand split (round 6).
//
//   C = act( alpha * (A (+ A2)) @ W^T + bias )      A: (M,K) fp32   W: (N,K) fp32, stored as three bf16 images
//
// Every fp32 number is EXACTLY the sum of three bf16 numbers: x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)
// (8 + 8 + 8 significant bits, round to nearest each time; bf16 has the fp32 exponent range).  A product is then nine bf16 x bf16
// products, each exact in fp32; the six largest -- hh, hm, mh, mm, hl, lh -- are accumulated by v_mfma_f32_32x32x16_bf16 with
// its fp32 accumulator, the three smallest (ml, lm: 2^-24, ll: 2^-32 of the product) are dropped: the error per product is of the
// size of one fp32 rounding, and there is no 256-term fp32 accumulation of rounded partial sums in front of it.  Six MFMAs of 32
// cycles per 16 k against eight fp32 MFMAs of 64 cycles (v_mfma_f32_32x32x2_f32): 0.375 of the fp32 matrix time.
//
// W is split once (roitr_split_bf16x3, at roitr_engine_finalize), A while it is staged: a thread rounds its 8 floats of the slab
// three times (v_cvt_pk_bf16_f32, two subtractions per piece) and writes one 16-byte fragment per piece.  Tile 64 x 64 TN, 4 waves
// (one 32 x 32 TN strip each), BK = 32; three row-major bf16 LDS images per operand with an 80-byte row pitch (the 16-byte fragment
// reads of 16 consecutive rows fall on 16 distinct 4-bank groups: 5 i mod 16).  The accumulation order of an output element --
// k in blocks of 16, ascending; per block lh, hl, mm, mh, hm, hh -- does not depend on TN or on the row count: rows are bitwise
// independent of the batch they are computed in, whatever tile a launch picks (tests/test_stages_gpu.py).
// Same RoitrGemm contract as gemm.hip for what the K >= 256 layers of the path use: row gather (a_idx), zero rows, the addend A2,
// the K-concatenated operand A_cat (+ a_cat_idx), bias, alpha, ReLU, the LayerNorm epilogue at 64 / 128 / 256 columns.
#include "common.h"
#include "prof.h"
#include "roitr_engine.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int PITCH = 40;   // bf16 elements per LDS row (80 bytes)

__device__ __forceinline__ unsigned pack_bf16(float x, float y)   // low half = x; round to nearest even
{
    f32x2 v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// (x, y) -> packed pairs of the three bf16 pieces; the residuals are exact in fp32
__device__ __forceinline__ void split2(float x, float y, unsigned& h, unsigned& m, unsigned& l)
{
    h = pack_bf16(x, y);
    const float rx = x - __uint_as_float(h << 16), ry = y - __uint_as_float(h & 0xffff0000u);
    m = pack_bf16(rx, ry);
    l = pack_bf16(rx - __uint_as_float(m << 16), ry - __uint_as_float(m & 0xffff0000u));
}

// TM x TN accumulators of 32 x 32 per wave: the block tile is 64 TM x 64 TN
template <int TM, int TN, bool LN, bool HA2>
__global__ __launch_bounds__(256) void gemm_x3_kernel(RoitrGemm g, int nx, int ny, int T)
{
    constexpr int TBN = BN * TN, TBM = BM * TM;
    constexpr int RP = TN == 4 ? 32 : 64;   // rows parked per LayerNorm pass
    constexpr int IMG_A = TBM * PITCH, IMG_B = TBN * PITCH;          // bf16 elements per piece image
    constexpr int STAGE_BYTES = 3 * (IMG_A + IMG_B) * 2;
    constexpr int TILE_BYTES = LN ? RP * (TBN + 1) * 4 : 4 * 32 * 36 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[STAGE_BYTES > TILE_BYTES ? STAGE_BYTES : TILE_BYTES];
    unsigned short* As = reinterpret_cast<unsigned short*>(smem_raw);   // [piece][row][PITCH]
    unsigned short* Bs = As + 3 * IMG_A;
    const int tile = xcd_block_id(T);
    if (tile >= T) return;
    const int by_ = tile / nx, bx_ = tile - by_ * nx;
    const float* A = g.A;
    const unsigned short* W = reinterpret_cast<const unsigned short*>(g.W);
    const float* bias = g.bias;
    float* C = g.C;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by_ * TBM, n0 = bx_ * TBN;
    const int r = tid >> 2, kq = (tid & 3) * 8;   // staging: rows r + 64 t, 8 consecutive k from kq of the 32-k slab

    const float* arow[TM]; const float* arow2[TM]; const float* arowc[TM]; const unsigned short* wrow[TN];
    {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            arow[t] = arow2[t] = arowc[t] = nullptr;
            const int am = m0 + r + 64 * t;
            if (am < g.M) {
                const int src = g.a_idx ? g.a_idx[am] : am;
                if (src >= 0 && (g.a_limit <= 0 || src < g.a_limit)) {
                    arow[t] = A + (size_t)src * g.lda;
                    if (HA2) arow2[t] = g.A2 + (size_t)src * g.lda;
                    if (g.A_cat) arowc[t] = g.A_cat + (size_t)(g.a_cat_idx ? g.a_cat_idx[am] : src) * g.lda_cat;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < TN; ++v) {
            const int wn_ = n0 + r + 64 * v;
            wrow[v] = wn_ < g.N ? W + (size_t)wn_ * g.ldw : nullptr;
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int v = 0; v < TN; ++v)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][v][i] = 0.f;

    float4 a0v[TM], a1v[TM], b0v[TM], b1v[TM];   // A (and A2) floats kq .. kq + 7 of the slab
    uint4 wv[TN][3];                             // the three pieces of W, 8 bf16 each
    auto fetch = [&](int k) {
        const bool cat = g.A_cat && k >= g.k_cat;   // slab-uniform: k_cat % 32 == 0
        const int kk = cat ? k - g.k_cat : k;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const float* p = cat ? arowc[t] : arow[t];
            if (p) { a0v[t] = *reinterpret_cast<const float4*>(p + kk); a1v[t] = *reinterpret_cast<const float4*>(p + kk + 4); }
            else a0v[t] = a1v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (HA2) {   // the addend covers the A part only
                if (arow2[t] && !cat) { b0v[t] = *reinterpret_cast<const float4*>(arow2[t] + k); b1v[t] = *reinterpret_cast<const float4*>(arow2[t] + k + 4); }
                else b0v[t] = b1v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int v = 0; v < TN; ++v)
#pragma unroll
            for (int p3 = 0; p3 < 3; ++p3)
                wv[v][p3] = wrow[v] ? *reinterpret_cast<const uint4*>(wrow[v] + (size_t)p3 * g.w_piece + k) : make_uint4(0, 0, 0, 0);
    };
    fetch(kq);
    const int kh = lane >> 5, ml = lane & 31;
    const unsigned short* ar = As + (wm * 32 * TM + ml) * PITCH + kh * 8;
    const unsigned short* br = Bs + (wn * 32 * TN + ml) * PITCH + kh * 8;
    unsigned short* aw = As + r * PITCH + kq;
    unsigned short* bw = Bs + r * PITCH + kq;
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            float4 x0 = a0v[t], x1 = a1v[t];
            if (HA2) { x0.x += b0v[t].x; x0.y += b0v[t].y; x0.z += b0v[t].z; x0.w += b0v[t].w; x1.x += b1v[t].x; x1.y += b1v[t].y; x1.z += b1v[t].z; x1.w += b1v[t].w; }
            uint4 h, m, l;
            split2(x0.x, x0.y, h.x, m.x, l.x); split2(x0.z, x0.w, h.y, m.y, l.y);
            split2(x1.x, x1.y, h.z, m.z, l.z); split2(x1.z, x1.w, h.w, m.w, l.w);
            unsigned short* a_ = aw + t * 64 * PITCH;
            *reinterpret_cast<uint4*>(a_) = h; *reinterpret_cast<uint4*>(a_ + IMG_A) = m; *reinterpret_cast<uint4*>(a_ + 2 * IMG_A) = l;
        }
#pragma unroll
        for (int v = 0; v < TN; ++v)
#pragma unroll
            for (int p3 = 0; p3 < 3; ++p3) *reinterpret_cast<uint4*>(bw + p3 * IMG_B + v * 64 * PITCH) = wv[v][p3];
        __syncthreads();
        if (k0 + BK < g.K) fetch(k0 + BK + kq);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 ah[TM], am[TM], al[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                ah[t] = *reinterpret_cast<const bf16x8*>(ar + t * 32 * PITCH + kk * 16);
                am[t] = *reinterpret_cast<const bf16x8*>(ar + IMG_A + t * 32 * PITCH + kk * 16);
                al[t] = *reinterpret_cast<const bf16x8*>(ar + 2 * IMG_A + t * 32 * PITCH + kk * 16);
            }
#pragma unroll
            for (int v = 0; v < TN; ++v) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(br + v * 32 * PITCH + kk * 16);
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(br + IMG_B + v * 32 * PITCH + kk * 16);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(br + 2 * IMG_B + v * 32 * PITCH + kk * 16);
#pragma unroll
                for (int t = 0; t < TM; ++t) {   // smallest terms first
                    acc[t][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], bh, acc[t][v], 0, 0, 0);
                    acc[t][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bl, acc[t][v], 0, 0, 0);
                    acc[t][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t], bm, acc[t][v], 0, 0, 0);
                    acc[t][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t], bh, acc[t][v], 0, 0, 0);
                    acc[t][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bm, acc[t][v], 0, 0, 0);
                    acc[t][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bh, acc[t][v], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // consumers of the prefetched registers stay below the MFMAs
    }
    if (LN) {   // the arithmetic of gemm.hip's epilogue / add_layernorm_kernel<TN>
        float* tile_ = reinterpret_cast<float*>(smem_raw);   // [RP][TBN + 1]
        float gam[TN], bet[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) { gam[i] = g.ln_gamma[lane + 64 * i]; bet[i] = g.ln_beta[lane + 64 * i]; }
        // a pass parks RP consecutive rows of the block: RP = 64: rows of both wave rows (TM == 1 only); RP = 32: one 32-row block
        static_assert(!LN || RP == 32 || TM == 1, "the 64-row pass assumes one row block per wave");
        for (int pass = 0; pass < TBM / RP; ++pass) {
            __syncthreads();   // every wave is done with the operand images / the previous pass
#pragma unroll
            for (int t = 0; t < TM; ++t)
            if (RP == BM ? true : (wm * TM + t == pass)) {
#pragma unroll
                for (int v = 0; v < TN; ++v) {
                    const int col = (wn * TN + v) * 32 + (lane & 31);
                    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int rl = (RP == BM ? wm * 32 : 0) + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                        tile_[rl * (TBN + 1) + col] = acc[t][v][i] * g.alpha + bv;
                    }
                }
            }
            __syncthreads();
            for (int rl = wave; rl < RP; rl += 4) {
                const int row = m0 + pass * RP + rl;
                if (row >= g.M) break;   // wave-uniform
                const float* rr = g.ln_res ? g.ln_res + (size_t)(g.ln_res_idx ? g.ln_res_idx[row] : row) * TBN : nullptr;
                float t[TN];
                float s_ = 0.f;
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    t[i] = tile_[rl * (TBN + 1) + lane + 64 * i];
                    if (rr) t[i] += rr[lane + 64 * i];
                    s_ += t[i];
                }
                const float mean = wave_sum(s_) / (float)TBN;
                float q_ = 0.f;
#pragma unroll
                for (int i = 0; i < TN; ++i) { const float d = t[i] - mean; q_ += d * d; }
                const float rstd = 1.0f / sqrtf(wave_sum(q_) / (float)TBN + g.ln_eps);
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    float y = (t[i] - mean) * rstd * gam[i] + bet[i];
                    if (g.ln_post) y += g.ln_post[(size_t)row * TBN + lane + 64 * i];
                    if (g.ln_relu) y = fmaxf(y, 0.f);
                    C[(size_t)row * g.ldc + lane + 64 * i] = y;
                }
            }
        }
        return;
    }
    // plain epilogue: a wave's 32 x 32 blocks go through its own 32 x 36 LDS scratch and leave as 16-byte stores (8 lanes per row)
    const bool wide = (g.ldc & 3) == 0 && (((uintptr_t)g.C) & 15) == 0;
    __syncthreads();   // the operand images are dead
    float* sc = reinterpret_cast<float*>(smem_raw) + wave * 32 * 36;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int v = 0; v < TN; ++v) {
        const int cb = n0 + (wn * TN + v) * 32;   // first column of this block
        const int rb = m0 + (wm * TM + t) * 32;   // first row
        if (cb >= g.N || rb >= g.M) continue;     // wave-uniform
        const int col = cb + (lane & 31);
        const float bv = (bias && col < g.N) ? bias[col] : 0.f;
        if (wide && cb + 32 <= g.N) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                float x = acc[t][v][i] * g.alpha + bv;
                if (g.relu) x = fmaxf(x, 0.f);
                sc[rl * 36 + (lane & 31)] = x;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int c4 = (lane & 7) * 4;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int rl = pass * 8 + (lane >> 3);
                const int row = rb + rl;
                if (row < g.M) *reinterpret_cast<float4*>(C + (size_t)row * g.ldc + cb + c4) = *reinterpret_cast<const float4*>(sc + rl * 36 + c4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the scratch is reused by the next block
        } else if (col < g.N) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = rb + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                if (row < g.M) {
                    float x = acc[t][v][i] * g.alpha + bv;
                    if (g.relu) x = fmaxf(x, 0.f);
                    C[(size_t)row * g.ldc + col] = x;
                }
            }
        }
    }
}

__global__ void split_bf16x3_kernel(long n, const float* __restrict__ src, unsigned short* __restrict__ dst, long piece)
{
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    const float x = src[i], y = i + 1 < n ? src[i + 1] : 0.f;
    unsigned h, m, l;
    split2(x, y, h, m, l);
    if (i + 1 < n) {
        *reinterpret_cast<unsigned*>(dst + i) = h; *reinterpret_cast<unsigned*>(dst + piece + i) = m; *reinterpret_cast<unsigned*>(dst + 2 * piece + i) = l;
    } else {
        dst[i] = (unsigned short)(h & 0xffffu); dst[piece + i] = (unsigned short)(m & 0xffffu); dst[2 * piece + i] = (unsigned short)(l & 0xffffu);
    }
}

}  // namespace

/* src (n fp32) -> three bf16 images of n elements each at dst, dst + piece, dst + 2 * piece (piece >= n, even; dst 4-byte aligned):
 * src[i] == h[i] + m[i] + l[i] exactly (up to underflow of the last piece). */
extern "C" int roitr_split_bf16x3(long n, const float* src, unsigned short* dst, long piece, hipStream_t stream)
{
    if (n <= 0) return ROITR_OK;
    if (((uintptr_t)dst & 3) != 0 || piece < n || (piece & 1)) return ROITR_ERR_ARG;
    split_bf16x3_kernel<<<(unsigned)((n / 2 + 1 + 255) / 256), 256, 0, stream>>>(n, src, dst, piece);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

/* shapes the split kernel takes (the engine asks before it gives a layer split weights) */
extern "C" int roitr_gemm_x3_supported(const RoitrGemm* g)
{
    if (!(g->bf16 & ROITR_BF16_X3) || (g->bf16 & ~ROITR_BF16_X3)) return 0;
    if (g->batch != 1 || g->seg_off || g->w_idx || g->ip_feat) return 0;
    if (g->K <= 0 || g->K % BK || g->lda % 4 || g->ldw % 8 || g->w_piece % 8 || g->w_piece < (long)g->N * g->ldw) return 0;
    if (((uintptr_t)g->A & 15) || ((uintptr_t)g->W & 15) || (g->A2 && ((uintptr_t)g->A2 & 15))) return 0;
    if (g->A_cat && (g->k_cat % BK || g->k_cat <= 0 || g->k_cat >= g->K || g->lda_cat % 4 || ((uintptr_t)g->A_cat & 15))) return 0;
    if (g->a_cat_idx && !g->A_cat) return 0;
    if (g->ln_gamma) {
        const int tn = g->N / BN;
        if (g->N % BN || (tn != 1 && tn != 2 && tn != 4) || g->relu || !g->ln_beta) return 0;
    }
    return 1;
}

int roitr_gemm_x3_launch(const RoitrGemm* g, hipStream_t stream)
{
    if (!roitr_gemm_x3_supported(g)) {
        roitr_set_error("roitr_gemm: shape / layout not supported by the bf16x3 kernel (K % 32, 16-byte rows, split weights, no batching)", __FILE__, __LINE__);
        return ROITR_ERR_UNSUPPORTED;
    }
    // tile: whole rows for the LayerNorm epilogue (64 x N); otherwise 64 x 256 where N allows and the grid fills the chip (measured at
    // M = 319 488, K = 256, N = 768: 64 x 64 73, 64 x 128 103, 64 x 256 131, 128 x 64 75, 128 x 128 101, 128 x 256 98 TFLOP/s: an A row
    // costs more than a W row -- fp32 loads, the split, and two of the four waves repeat none of it) down to 64 x 64 on small grids
    // (the result bits are the same whatever the tile)
    int tm = 1, tn = 1;
    if (g->ln_gamma) tn = g->N / BN;
    else {
        tn = g->N % 256 == 0 ? 4 : (g->N % 128 == 0 ? 2 : 1);
        while (tm * tn > 1 && (long)div_up(g->M, BM * tm) * div_up(g->N, BN * tn) < 512) { if (tm > 1) tm >>= 1; else tn >>= 1; }
    }
    const int ny = div_up(g->M, BM * tm), nx = div_up(g->N, BN * tn);
    const long Tl = (long)nx * ny;
    if (Tl > 0x7ffffff0L) return ROITR_ERR_UNSUPPORTED;
    const int T = (int)Tl;
    const unsigned grid = (unsigned)xcd_grid(T);
    const int prof_cls = roitr_prof_is_enabled() ? roitr_gemm_prof_class(g) : ROITR_PROF_GEMM;
    roitr_prof_begin2(prof_cls, 2.0 * g->M * g->N * (double)g->K, roitr_gemm_algorithmic_bytes(g), stream);
    const bool a2 = g->A2 != nullptr;
#define X3_LAUNCH(TM_, TN_, LN_) \
    do { if (a2) gemm_x3_kernel<TM_, TN_, LN_, true><<<grid, 256, 0, stream>>>(*g, nx, ny, T); \
         else gemm_x3_kernel<TM_, TN_, LN_, false><<<grid, 256, 0, stream>>>(*g, nx, ny, T); } while (0)
    if (g->ln_gamma) { if (tn == 1) X3_LAUNCH(1, 1, true); else if (tn == 2) X3_LAUNCH(1, 2, true); else X3_LAUNCH(1, 4, true); }
    else if (tm == 2 && tn == 2) X3_LAUNCH(2, 2, false);
    else if (tm == 2 && tn == 1) X3_LAUNCH(2, 1, false);
    else if (tm == 1 && tn == 2) X3_LAUNCH(1, 2, false);
    else if (tm == 1 && tn == 4) X3_LAUNCH(1, 4, false);
    else if (tm == 2 && tn == 4) X3_LAUNCH(2, 4, false);
    else X3_LAUNCH(1, 1, false);
#undef X3_LAUNCH
    roitr_prof_end(prof_cls, stream);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}
