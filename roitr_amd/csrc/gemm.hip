// fp32 linear layers on the gfx950 matrix cores.
//
// C[b] = act( alpha * (A[b] (+ A2[b])) @ W[b]^T + bias[b] )      A: (M,K)  W: (N,K) (torch Linear layout)
//
// Every dense layer of the RoITr path goes through this one kernel (reference: the nn.Linear calls in
// model/transformer/*.py, model/model.py, model/RIGA_v2.py:64-68, and the einsum of RIGA_v2.py:150).
// v_mfma_f32_32x32x2_f32: exact fp32 FMA chains in k order (bit-reproducible, no TF32-style
// truncation exists on gfx950), 64 FLOP/clk/SIMD.  64x64 block tile, 4 waves (one 32x32 MFMA tile
// each), BK = 32.  The 32x32x2 MFMA wants, per lane (m = lane&31, kh = lane>>5), the operand element
// A[m][2*kk + kh] for kk = 0..15: the LDS image is therefore [kh][row][kk] (even / odd k planes, 16 kk contiguous,
// row pitch 20 floats) so that a lane fetches its 16 operands of a K-slab with four conflict-free ds_read_b128 and
// the stager writes its 8 consecutive k as two ds_write_b128.  All 32 operands of a slab are in VGPRs before the
// 16 back-to-back MFMAs start (no LDS latency inside the dependent accumulator chain -- that chain, not HBM, is
// what bounds the many small-M launches of this path); the next K-slab is prefetched from HBM/L2 into registers
// while the current one feeds the MFMAs.
//
// Row gathers on either operand (with "index >= limit -> zero row", which is how the reference's
// padded patches are built, RIGA_v2.py:129-142) and an elementwise addend on A (x + pos of the
// cross-attention, geoattention.py:44-45) are fused into the staging loads.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include "prof.h"
#include "roitr_engine.h"

#ifndef GEMM_WIDE_STORE
#define GEMM_WIDE_STORE 1
#endif
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 32, LDR = 20;  // LDR: row pitch (floats) of the [kh][row][kk] LDS image

// 8 consecutive k of one row into registers (zero row when p == nullptr; scalar tail when unaligned / past K).
// No arithmetic here: the consumer of these registers is the NEXT iteration's LDS write, so the loads stay in
// flight across the MFMA block.
__device__ __forceinline__ void load8(const float* __restrict__ p, int k, int K, bool vec_ok, float (&v)[8])
{
    if (p == nullptr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        return;
    }
    if (vec_ok && k + 8 <= K) {
        const float4 a = *reinterpret_cast<const float4*>(p + k);
        const float4 b = *reinterpret_cast<const float4*>(p + k + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (k + i < K) ? p[k + i] : 0.f;
    }
}

// FAST: K % 32 == 0 (<= ZERO_ROW_LEN) and 16-byte aligned rows -> the staging loads are unconditional float4 loads
// (rows that do not exist are redirected to a resident all-zero row), and a scheduling barrier after the MFMA block
// keeps every use of the prefetched registers behind it, so nothing forces a vmcnt wait between issuing the
// prefetch and the MFMAs.  The generic variant keeps bounds-checked scalar tails.
constexpr int ZERO_ROW_LEN = 2048;
constexpr int GEMM_SMALL_MAX_TILES = 512;  // 64 x 64 tiles below which gemm_small_kernel (32 x 32 tiles, v_mfma_f32_16x16x4_f32) takes the launch
constexpr int GEMM_IL_MIN_TILES = 1024;   // below: a few tiles per CU, the one accumulator chain of a wave is the critical path (one-pair-per-call launches: <= 628 tiles)
__device__ __attribute__((aligned(16))) float g_zero_row[ZERO_ROW_LEN];

// TN: 32x32 accumulators per wave along N; the block tile is 64 x (64 TN).  TN = 1 is the GEMM of the path (see the
// launcher for why wider tiles are not used for plain GEMMs); TN = 2 / 4 exist for the LayerNorm epilogue, which needs
// whole rows (N = 128 / 256) inside one block.
// LN: fused LayerNorm epilogue (RoitrGemm::ln_*), N == 64 TN: the finished tile is parked row-major in the staging LDS
// and every wave normalises 16 full rows with exactly the arithmetic (and summation order) of add_layernorm_kernel<TN>,
// so the result is bitwise that of the two-launch sequence while the (M, N) intermediate never touches HBM.
// HA2 (FAST only): the launch has an elementwise addend A2 -- a template parameter, because a run-time `if (A2)` around staging
// loads that are spread over the MFMA stream would turn their registers into phi nodes (hipcc then copies behind the loads and waits).
template <bool FAST, int TN, bool LN, bool HA2 = false, bool IL = false>
__global__ __launch_bounds__(256) void gemm_kernel(RoitrGemm g, int nx, int ny, int T)
{
    constexpr bool WIDE_STORE = GEMM_WIDE_STORE != 0;
    constexpr int TBN = BN * TN;
    constexpr int RP = TN == 4 ? 32 : 64;   // rows parked per LayerNorm pass (keeps the static LDS under 64 KB at TN = 4)
    constexpr int STAGE = 2 * BM * LDR + 2 * TBN * LDR, TILE = LN ? RP * (TBN + 1) : 0;
    __shared__ __attribute__((aligned(16))) float smem[STAGE > TILE ? STAGE : TILE];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDR;
    // 1-D XCD-aware tile grid: XCD x = blockIdx % 8 gets the contiguous tile range [x T/8, (x+1) T/8), N tiles of one
    // row block adjacent, so the A rows of a row block are fetched into ONE L2 instead of nx different ones
    if (g.batch_live) {   // batch list with a device-side live length: the tile map covers the live tiles only (all eight XCDs stay busy)
        const long tl = (long)*g.batch_live * nx * ny;
        if (tl < T) T = (int)tl;
    }
    const int tile = xcd_block_id_live(T);
    if (tile < 0) return;
    const int bz = tile / (nx * ny);
    const int rem = tile - bz * nx * ny;
    const int by_ = rem / nx, bx_ = rem - by_ * nx;
    const float* A = g.A + (size_t)bz * g.sA;
    const float* A2 = g.A2 ? g.A2 + (size_t)bz * g.sA : nullptr;
    const float* W = g.W + (size_t)bz * g.sW;
    const float* bias = g.bias ? g.bias + (size_t)bz * g.sBias : nullptr;
    float* C = g.C + (size_t)bz * g.sC;
    const int* a_idx = g.a_idx ? g.a_idx + (size_t)bz * g.sAidx : nullptr;
    const int* w_idx = g.w_idx ? g.w_idx + (size_t)bz * g.sWidx : nullptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by_ * BM, n0 = bx_ * TBN;
    constexpr bool SPLIT16 = FAST;
    const int r = tid >> 2, kq = (tid & 3) * 8;
    const int kf = SPLIT16 ? (tid & 3) * 4 : kq;   // staging-load offset inside the slab
    if (g.seg_off) {  // ragged batch: this batch's row segments of A and W
        const int ia = g.seg_a0 + bz, iw = g.seg_w0 + bz;
        const int a0 = ia == 0 ? 0 : g.seg_off[ia - 1], w0 = iw == 0 ? 0 : g.seg_off[iw - 1];
        g.M = g.seg_off[ia] - a0; g.N = g.seg_off[iw] - w0;
        A += (size_t)a0 * g.lda; W += (size_t)w0 * g.ldw;
        if (A2) A2 += (size_t)a0 * g.lda;
        if (m0 >= g.M || n0 >= g.N) return;  // block-uniform
    }

    const float* arow = nullptr; const float* arow2 = nullptr; const float* arowc = nullptr; const float* wrow[TN];
    {
        const int am = m0 + r;
        if (am < g.M) {
            int src = a_idx ? a_idx[am] : am;
            if (src >= 0 && (g.a_limit <= 0 || src < g.a_limit)) {
                arow = A + (size_t)src * g.lda;
                if (A2) arow2 = A2 + (size_t)src * g.lda;
                // column k >= k_cat of the product = A_cat[k - k_cat]; its row: the A row's, or its own gather (a_cat_idx)
                if (FAST && g.A_cat) arowc = g.A_cat + (size_t)(g.a_cat_idx ? g.a_cat_idx[am] : src) * g.lda_cat;
            }
        }
#pragma unroll
        for (int v = 0; v < TN; ++v) {
            wrow[v] = nullptr;
            const int wn_ = n0 + r + 64 * v;
            if (wn_ < g.N) {
                int src = w_idx ? w_idx[wn_] : wn_;
                if (src >= 0 && (g.w_limit <= 0 || src < g.w_limit)) wrow[v] = W + (size_t)src * g.ldw;
            }
        }
    }
    const bool a_vec = (g.lda % 4 == 0) && (((uintptr_t)A & 15) == 0) && (!A2 || ((uintptr_t)A2 & 15) == 0);
    const bool w_vec = (g.ldw % 4 == 0) && (((uintptr_t)W & 15) == 0);

    f32x16 acc[TN];
#pragma unroll
    for (int v = 0; v < TN; ++v)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[v][i] = 0.f;

    float av[8], a2v[8], wv[TN][8];
    if (FAST) {
        if (!arow) arow = g_zero_row;
        if (!arow2) arow2 = g_zero_row;
        if (g.A_cat && !arowc) arowc = g_zero_row;
#pragma unroll
        for (int v = 0; v < TN; ++v) if (!wrow[v]) wrow[v] = g_zero_row;
    }
    // FAST: the four lanes of a row read the slab as two fully used 64-byte segments (lane j: floats 4j.. and 16+4j..), not
    // as interleaved 16-byte pieces of both; which 8 k of the slab a lane stages is free as long as A and W agree.
    auto ld8 = [&](const float* p, float (&d)[8]) {
        const float4 x = *reinterpret_cast<const float4*>(p), y = *reinterpret_cast<const float4*>(p + (SPLIT16 ? 16 : 4));
        d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w; d[4] = y.x; d[5] = y.y; d[6] = y.z; d[7] = y.w;
    };
    auto fetch = [&](int k) {
        if (FAST) {
            ld8((g.A_cat && k >= g.k_cat) ? arowc + (k - g.k_cat) : arow + k, av);   // slab-uniform: k_cat % 32 == 0
#pragma unroll
            for (int v = 0; v < TN; ++v) ld8(wrow[v] + k, wv[v]);
            if (HA2) ld8((g.A_cat && k >= g.k_cat) ? g_zero_row + (k & 31) : arow2 + k, a2v);   // the addend covers the A part only
        } else {
            load8(arow, k, g.K, a_vec, av);
            load8(arow2, k, g.K, a_vec, a2v);
#pragma unroll
            for (int v = 0; v < TN; ++v) load8(wrow[v], k, g.K, w_vec, wv[v]);
        }
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) a2v[i] = 0.f;
    fetch(kf);
    const int kh = lane >> 5, ml = lane & 31;
    const float4* ar = reinterpret_cast<const float4*>(As + (kh * BM + wm * 32 + ml) * LDR);
    const float4* br = reinterpret_cast<const float4*>(Bs + (kh * TBN + wn * 32 * TN + ml) * LDR);
    float4* aw0 = reinterpret_cast<float4*>(As + (0 * BM + r) * LDR + (kq >> 1));
    float4* aw1 = reinterpret_cast<float4*>(As + (1 * BM + r) * LDR + (kq >> 1));
    float4* bw0 = reinterpret_cast<float4*>(Bs + (0 * TBN + r) * LDR + (kq >> 1));
    float4* bw1 = reinterpret_cast<float4*>(Bs + (1 * TBN + r) * LDR + (kq >> 1));
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        __syncthreads();
        if (FAST ? HA2 : true) {
#pragma unroll
            for (int i = 0; i < 8; ++i) av[i] += a2v[i];
        }
        *aw0 = make_float4(av[0], av[2], av[4], av[6]); *aw1 = make_float4(av[1], av[3], av[5], av[7]);
#pragma unroll
        for (int v = 0; v < TN; ++v) {
            bw0[v * 64 * LDR / 4] = make_float4(wv[v][0], wv[v][2], wv[v][4], wv[v][6]);
            bw1[v * 64 * LDR / 4] = make_float4(wv[v][1], wv[v][3], wv[v][5], wv[v][7]);
        }
        __syncthreads();
        if (FAST && IL) {
            // Round 4: the staging loads of the next slab are ISSUED ONE AT A TIME BETWEEN THE MFMAs of this one.  Measured
            // (scripts/micro/gemm_gen2.hip): with the 4 loads issued back to back behind the barrier the kernel runs at 99-113
            // TFLOP/s whether the data is cache-hot or not and whether anybody waits for it or not; without them at 121-129 -- a
            // VMEM instruction occupies its wave's issue slot until the address path accepts it, and four waves of a block (one
            // per SIMD) present theirs in the same cycles.  Spread over the 16 TN MFMAs every issue hides under a running MFMA
            // (+7-13 % on the K >= 256 shapes alone, gemm family 38.3 -> 36.9 ms per 512-pair step in the forward; bitwise the same
            // results).  All fragments of the slab are read first, so nothing but MFMAs and these loads remains in the stream.  The
            // last slab re-fetches itself (no branch in the stream).  IL = false (grids below GEMM_IL_MIN_TILES: the one-pair-per-call
            // launches, a few waves on the whole chip) keeps the burst: there a load between two MFMAs of the ONE accumulator chain
            // a wave has only lengthens the chain (3.48 -> 3.60 ms per pair with the interleave everywhere).
            constexpr bool UPFRONT = TN <= 2;   // TN = 4: 80 fragment registers would cost the third wave per SIMD -- read per quarter
            float4 af[4], bf[TN][4];
            if (UPFRONT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    af[q] = ar[q];
#pragma unroll
                    for (int v = 0; v < TN; ++v) bf[v][q] = br[v * 32 * LDR / 4 + q];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const int kn = (k0 + BK < g.K ? k0 + BK : k0) + kf;
            const float* an = (g.A_cat && kn >= g.k_cat) ? arowc + (kn - g.k_cat) : arow + kn;   // slab-uniform: k_cat % 32 == 0
            constexpr int NS = 2 * (1 + (HA2 ? 1 : 0) + TN), TOT = 16 * TN;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!UPFRONT) {
                    af[q] = ar[q];
#pragma unroll
                    for (int v = 0; v < TN; ++v) bf[v][q] = br[v * 32 * LDR / 4 + q];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int v = 0; v < TN; ++v) {
                        const float a = c == 0 ? af[q].x : c == 1 ? af[q].y : c == 2 ? af[q].z : af[q].w;
                        const float b = c == 0 ? bf[v][q].x : c == 1 ? bf[v][q].y : c == 2 ? bf[v][q].z : bf[v][q].w;
                        acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[v], 0, 0, 0);
                        const int n = (q * 4 + c) * TN + v;
#pragma unroll
                        for (int i = 0; i < NS; ++i)
                            if (n == (i * TOT + TOT / 2) / NS) {
                                __builtin_amdgcn_sched_barrier(0);
                                const int h = i & 1, o = h ? (SPLIT16 ? 16 : 4) : 0, j = i >> 1;   // half h of staging row j
                                const float* src = j == 0 ? an : (HA2 && j == 1) ? ((g.A_cat && kn >= g.k_cat) ? g_zero_row + (kn & 31) : arow2 + kn)
                                                                  : wrow[j - 1 - (HA2 ? 1 : 0)] + kn;
                                const float4 x = *reinterpret_cast<const float4*>(src + o);
                                float* dst = j == 0 ? av : (HA2 && j == 1) ? a2v : wv[j - 1 - (HA2 ? 1 : 0)];
                                dst[4 * h] = x.x; dst[4 * h + 1] = x.y; dst[4 * h + 2] = x.z; dst[4 * h + 3] = x.w;
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);  // consumers of the prefetched registers stay below the MFMAs
        } else {
            if (k0 + BK < g.K) fetch(k0 + BK + kf);
            if (FAST && TN == 1) {   // every operand of the slab in VGPRs before the 16 back-to-back MFMAs
                float4 af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { af[i] = ar[i]; bf[i] = br[i]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[i].x, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[i].y, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[i].z, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[i].w, acc[0], 0, 0, 0);
                }
            } else
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 af = ar[q];
                float4 bf[TN];
#pragma unroll
                for (int v = 0; v < TN; ++v) bf[v] = br[v * 32 * LDR / 4 + q];
#pragma unroll
                for (int v = 0; v < TN; ++v) {
                    acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf[v].x, acc[v], 0, 0, 0);
                    acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf[v].y, acc[v], 0, 0, 0);
                    acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf[v].z, acc[v], 0, 0, 0);
                    acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf[v].w, acc[v], 0, 0, 0);
                }
            }
            if (FAST) __builtin_amdgcn_sched_barrier(0);  // consumers of the prefetched registers stay below the MFMAs
        }
    }
    if (LN) {
        float* tile_ = smem;   // [RP][TBN + 1]
        float gam[TN], bet[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) { gam[i] = g.ln_gamma[lane + 64 * i]; bet[i] = g.ln_beta[lane + 64 * i]; }
        for (int pass = 0; pass < BM / RP; ++pass) {
            __syncthreads();   // every wave is done with the operand images / the previous pass
            if (RP == BM || wm == pass) {
#pragma unroll
                for (int v = 0; v < TN; ++v) {
                    const int col = (wn * TN + v) * 32 + (lane & 31);
                    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int rl = (RP == BM ? wm * 32 : 0) + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                        tile_[rl * (TBN + 1) + col] = acc[v][i] * g.alpha + bv;
                    }
                }
            }
            __syncthreads();
            // TransitionUp addend: lane t holds the descriptors of the wave's t-th row of this pass (indices, normalised weights:
            // interp3_add_kernel's expressions), the rows' feature values are fetched one row AHEAD of their use
            int pi0 = 0, pi1 = 0, pi2 = 0; float pw0 = 0.f, pw1 = 0.f, pw2 = 0.f;
            float fn0[TN], fn1[TN], fn2[TN];
            if (g.ip_feat) {
                const int rowl = m0 + pass * RP + wave + 4 * lane;
                if (lane < RP / 4 && rowl < g.M) {
                    const int* ii = g.ip_idx + (size_t)rowl * 3;
                    const float* dd = g.ip_dist2 + (size_t)rowl * 3;
                    pi0 = ii[0]; pi1 = ii[1]; pi2 = ii[2];
                    pw0 = 1.0f / (sqrtf(dd[0]) + 1e-8f); pw1 = 1.0f / (sqrtf(dd[1]) + 1e-8f); pw2 = 1.0f / (sqrtf(dd[2]) + 1e-8f);
                    const float ws = (pw0 + pw1) + pw2;
                    pw0 /= ws; pw1 /= ws; pw2 /= ws;
                }
                const float* a0 = g.ip_feat + (size_t)__builtin_amdgcn_readlane(pi0, 0) * TBN;
                const float* a1 = g.ip_feat + (size_t)__builtin_amdgcn_readlane(pi1, 0) * TBN;
                const float* a2 = g.ip_feat + (size_t)__builtin_amdgcn_readlane(pi2, 0) * TBN;
#pragma unroll
                for (int i = 0; i < TN; ++i) { fn0[i] = a0[lane + 64 * i]; fn1[i] = a1[lane + 64 * i]; fn2[i] = a2[lane + 64 * i]; }
            }
            // residual / post-add rows of the wave's next HB rows requested up front (round 5): inside the row loop every row waited
            // for its own round trip behind two wave reductions -- the LayerNorm launches of levels 1 - 2 ran at 2 TB/s.  Eight rows
            // at a time: all sixteen cost the 128-column kernel its fourth wave per SIMD (132 VGPRs).
            constexpr int RW = RP / 4, HB = TN == 4 ? 2 : (RW > 8 ? 8 : RW);
            float resv[HB][TN], postv[HB][TN];
#pragma unroll 1
            for (int u0 = 0; u0 < RW; u0 += HB) {
                if (g.ln_res || g.ln_post) {
#pragma unroll
                    for (int w = 0; w < HB; ++w) {
                        const int row_ = m0 + pass * RP + wave + 4 * (u0 + w);
#pragma unroll
                        for (int i = 0; i < TN; ++i) { resv[w][i] = 0.f; postv[w][i] = 0.f; }
                        if (row_ < g.M) {
                            if (g.ln_res) {
                                const float* rr = g.ln_res + (size_t)(g.ln_res_idx ? g.ln_res_idx[row_] : row_) * TBN;
#pragma unroll
                                for (int i = 0; i < TN; ++i) resv[w][i] = rr[lane + 64 * i];
                            }
                            if (g.ln_post) {
#pragma unroll
                                for (int i = 0; i < TN; ++i) postv[w][i] = g.ln_post[(size_t)row_ * TBN + lane + 64 * i];
                            }
                        }
                    }
                }
#pragma unroll
              for (int w_ = 0; w_ < HB; ++w_) {
                const int rl = wave + 4 * (u0 + w_);
                const int row = m0 + pass * RP + rl;
                if (row >= g.M) break;   // wave-uniform
                float fc0[TN], fc1[TN], fc2[TN];
                float w0 = 0.f, w1 = 0.f, w2 = 0.f;
                if (g.ip_feat) {
                    const int t_ = (rl - wave) >> 2;
#pragma unroll
                    for (int i = 0; i < TN; ++i) { fc0[i] = fn0[i]; fc1[i] = fn1[i]; fc2[i] = fn2[i]; }
                    w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pw0), t_));
                    w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pw1), t_));
                    w2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pw2), t_));
                    if (rl + 4 < RP && row + 4 < g.M) {        // the next row's feature rows: in flight across this row's LayerNorm
                        const float* a0 = g.ip_feat + (size_t)__builtin_amdgcn_readlane(pi0, t_ + 1) * TBN;
                        const float* a1 = g.ip_feat + (size_t)__builtin_amdgcn_readlane(pi1, t_ + 1) * TBN;
                        const float* a2 = g.ip_feat + (size_t)__builtin_amdgcn_readlane(pi2, t_ + 1) * TBN;
#pragma unroll
                        for (int i = 0; i < TN; ++i) { fn0[i] = a0[lane + 64 * i]; fn1[i] = a1[lane + 64 * i]; fn2[i] = a2[lane + 64 * i]; }
                    }
                }
                float t[TN];
                float s_ = 0.f;
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    t[i] = tile_[rl * (TBN + 1) + lane + 64 * i];
                    if (g.ln_res) t[i] += resv[w_][i];
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
                    if (g.ln_post) y += postv[w_][i];
                    if (g.ln_relu) y = fmaxf(y, 0.f);
                    if (g.ip_feat) {   // TransitionUp: + three-nearest-neighbour interpolation, after the activation
                        float acc = 0.f;
                        acc += fc0[i] * w0; acc += fc1[i] * w1; acc += fc2[i] * w2;
                        y = y + acc;
                    }
                    C[(size_t)row * g.ldc + lane + 64 * i] = y;
                }
              }
            }
        }
        return;
    }
    if (TN == 1 && WIDE_STORE && n0 + BN <= g.N && (g.ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0) {
        // full 64-column tile: transpose through the staging LDS and store 16 bytes per lane (4 store instructions per
        // wave instead of 16 four-byte ones)
        constexpr int TP = BN + 4;   // row pitch: 16-byte aligned rows
        static_assert(BM * TP <= 2 * BM * LDR + 2 * TBN * LDR || TN != 1, "tile must fit the staging LDS");
        __syncthreads();
        float* tile_ = smem;
        {
            const int col = wn * 32 + (lane & 31);
            const float bv = bias ? bias[n0 + col] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                float x = acc[0][i] * g.alpha + bv;
                if (g.relu) x = fmaxf(x, 0.f);
                tile_[rl * TP + col] = x;
            }
        }
        __syncthreads();
        const int c4 = (tid & 15) * 4;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int rl = pass * 16 + (tid >> 4);
            const int row = m0 + rl;
            if (row < g.M) *reinterpret_cast<float4*>(C + (size_t)row * g.ldc + n0 + c4) = *reinterpret_cast<const float4*>(tile_ + rl * TP + c4);
        }
        return;
    }
#pragma unroll
    for (int v = 0; v < TN; ++v) {
        const int col = n0 + (wn * TN + v) * 32 + (lane & 31);
        if (col < g.N) {
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                if (row < g.M) {
                    float x = acc[v][i] * g.alpha + bv;
                    if (g.relu) x = fmaxf(x, 0.f);
                    C[(size_t)row * g.ldc + col] = x;
                }
            }
        }
    }
}


// ---------------------------------------------------------------- small grids: 32 x 32 tiles of v_mfma_f32_16x16x4_f32
// The launches of a one-pair (or few-pair) call have a few dozen 64 x 64 tiles for 1024 SIMDs: what bounds them is the ONE
// accumulator chain of a wave -- K / 2 dependent v_mfma_f32_32x32x2_f32 at 64 cycles each (3.4 us at K = 256) -- not the matrix
// pipe's throughput.  v_mfma_f32_16x16x4_f32 sums 4 k per instruction in 32 cycles: the same FLOP rate, a quarter of the chain
// for a quarter of the tile.  This kernel gives every wave a 16 x 16 tile (block = 32 x 32, four waves), i.e. spreads the same work
// over 4x the waves, each with a 4x shorter chain.  Both instructions are exact fp32 FMA chains in ascending k order
// (scripts/micro/mfma_korder.hip: bit-identical to a scalar fmaf loop, and therefore to each other), so a row's result does not
// depend on which kernel its launch was given (the k order inside a slab is gemm_kernel's, see `ks`):
// test_gemm_rows_do_not_depend_on_the_row_count covers grids on both sides of the switch.  Same operand options as gemm_kernel<FAST = true, TN = 1, LN = false>; LDS image row-major [32][36] per operand, a
// lane (i = lane & 15, g = lane >> 4) reads element k = 4 s + g of its row for step s (16 single-dword reads per 32-k slab, all
// before the 8 MFMAs of the slab).
constexpr int SM = 32, SLD = 36;
typedef float f32x4s __attribute__((ext_vector_type(4)));
template <bool HA2>
__global__ __launch_bounds__(256) void gemm_small_kernel(RoitrGemm g, int nx, int ny, int T)
{
    __shared__ __attribute__((aligned(16))) float As[SM * SLD];
    __shared__ __attribute__((aligned(16))) float Bs[SM * SLD];
    if (g.batch_live) {   // batch list with a device-side live length: the tile map covers the live tiles only (all eight XCDs stay busy)
        const long tl = (long)*g.batch_live * nx * ny;
        if (tl < T) T = (int)tl;
    }
    const int tile = xcd_block_id_live(T);
    if (tile < 0) return;
    const int bz = tile / (nx * ny);
    const int rem = tile - bz * nx * ny;
    const int by_ = rem / nx, bx_ = rem - by_ * nx;
    const float* A = g.A + (size_t)bz * g.sA;
    const float* A2 = g.A2 ? g.A2 + (size_t)bz * g.sA : nullptr;
    const float* W = g.W + (size_t)bz * g.sW;
    const float* bias = g.bias ? g.bias + (size_t)bz * g.sBias : nullptr;
    float* C = g.C + (size_t)bz * g.sC;
    const int* a_idx = g.a_idx ? g.a_idx + (size_t)bz * g.sAidx : nullptr;
    const int* w_idx = g.w_idx ? g.w_idx + (size_t)bz * g.sWidx : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by_ * SM, n0 = bx_ * SM;
    const int r = tid >> 3, kf = (tid & 7) * 4;
    if (g.seg_off) {  // ragged batch: this batch's row segments of A and W
        const int ia = g.seg_a0 + bz, iw = g.seg_w0 + bz;
        const int a0 = ia == 0 ? 0 : g.seg_off[ia - 1], w0 = iw == 0 ? 0 : g.seg_off[iw - 1];
        g.M = g.seg_off[ia] - a0; g.N = g.seg_off[iw] - w0;
        A += (size_t)a0 * g.lda; W += (size_t)w0 * g.ldw;
        if (A2) A2 += (size_t)a0 * g.lda;
        if (m0 >= g.M || n0 >= g.N) return;  // block-uniform
    }
    const float* arow = g_zero_row; const float* arow2 = g_zero_row; const float* arowc = g_zero_row; const float* wrow = g_zero_row;
    {
        const int am = m0 + r;
        if (am < g.M) {
            const int src = a_idx ? a_idx[am] : am;
            if (src >= 0 && (g.a_limit <= 0 || src < g.a_limit)) {
                arow = A + (size_t)src * g.lda;
                if (A2) arow2 = A2 + (size_t)src * g.lda;
                if (g.A_cat) arowc = g.A_cat + (size_t)(g.a_cat_idx ? g.a_cat_idx[am] : src) * g.lda_cat;
            }
        }
        const int wn_ = n0 + r;
        if (wn_ < g.N) {
            const int src = w_idx ? w_idx[wn_] : wn_;
            if (src >= 0 && (g.w_limit <= 0 || src < g.w_limit)) wrow = W + (size_t)src * g.ldw;
        }
    }
    f32x4s acc = {0.f, 0.f, 0.f, 0.f};
    float4 av, wv, a2v = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int k) {   // k: slab start + kf; the K-concatenated part is slab-uniform (k_cat % 32 == 0)
        const bool catp = g.A_cat && k >= g.k_cat;
        av = *reinterpret_cast<const float4*>(catp ? arowc + (k - g.k_cat) : arow + k);
        wv = *reinterpret_cast<const float4*>(wrow + k);
        if (HA2) a2v = *reinterpret_cast<const float4*>(catp ? g_zero_row + (k & 31) : arow2 + k);   // the addend covers the A part only
    };
    fetch(kf);
    const int i = lane & 15, gq = lane >> 4;
    const float* ar = As + (wm * 16 + i) * SLD + gq;
    const float* br = Bs + (wn * 16 + i) * SLD + gq;
    // gemm_kernel's summation order inside a 32-k slab is the order of ITS LDS image, whose stager puts global k 4j..4j+3 and
    // 16+4j..16+4j+3 into slots 8j..8j+7 (two fully used 64-byte segments per row); same slots here, or the two kernels round differently
    const int ks = (tid & 7) < 4 ? (tid & 7) * 8 : ((tid & 7) - 4) * 8 + 4;
    float4* aw = reinterpret_cast<float4*>(As + r * SLD + ks);
    float4* bw = reinterpret_cast<float4*>(Bs + r * SLD + ks);
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        __syncthreads();
        if (HA2) { av.x += a2v.x; av.y += a2v.y; av.z += a2v.z; av.w += a2v.w; }
        *aw = av; *bw = wv;
        __syncthreads();
        if (k0 + BK < g.K) fetch(k0 + BK + kf);
        float af[8], bf[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) { af[s] = ar[4 * s]; bf[s] = br[4 * s]; }
        __builtin_amdgcn_sched_barrier(0);   // one LDS round trip per slab: hipcc otherwise sinks the reads between the MFMAs, four waits per slab
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // consumers of the prefetched registers stay below the MFMAs
    }
    // 16x16 C/D map: col = lane & 15, row = 4 (lane >> 4) + reg
    const int col = n0 + wn * 16 + i;
    if (col < g.N) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = m0 + wm * 16 + 4 * gq + q;
            if (row < g.M) {
                float x = acc[q] * g.alpha + bv;
                if (g.relu) x = fmaxf(x, 0.f);
                C[(size_t)row * g.ldc + col] = x;
            }
        }
    }
}

}  // namespace

namespace {
struct ShapeStat { double ms = 0; long n = 0; };
struct ShapeLog {
    std::map<std::string, ShapeStat> m;
    ~ShapeLog() {
        for (auto& kv : m) fprintf(stderr, "GEMMSHAPE %s launches %ld ms %.4f\n", kv.first.c_str(), kv.second.n, kv.second.ms);
    }
};
void shape_log(const RoitrGemm* g, bool fast, float ms)
{
    static ShapeLog log;
    char key[160];
    snprintf(key, sizeof key, "M %d N %d K %d batch %d gather %d a2 %d relu %d seg %d fast %d ln %d bf16 %d", g->M, g->N, g->K, g->batch, g->a_idx != nullptr,
             g->A2 != nullptr, g->relu, g->seg_off != nullptr, (int)fast, g->ln_gamma != nullptr, g->bf16);
    auto& st = log.m[key];
    st.ms += ms; st.n += 1;
}
}  // namespace

int roitr_gemm_bf16_launch(const RoitrGemm* g, hipStream_t stream);   // gemm_bf16.hip
int roitr_gemm_x3_launch(const RoitrGemm* g, hipStream_t stream);     // gemm_x3.hip

extern "C" int roitr_gemm(const RoitrGemm* g, hipStream_t stream)
{
    if (g->M <= 0 || g->N <= 0 || g->batch <= 0) return ROITR_OK;
    if (g->K <= 0 || !g->A || !g->W || !g->C) { roitr_set_error("roitr_gemm: K <= 0 or a null operand", __FILE__, __LINE__); return ROITR_ERR_ARG; }
    if (g->bf16) {
        static const bool shapes_h = getenv("ROITR_GEMM_SHAPES") != nullptr;  // debug: per-shape timing table at exit (synchronous)
        hipEvent_t h0 = nullptr, h1 = nullptr;
        if (shapes_h) { hipEventCreate(&h0); hipEventCreate(&h1); hipEventRecord(h0, stream); }
        const int rc = (g->bf16 & ROITR_BF16_X3) ? roitr_gemm_x3_launch(g, stream) : roitr_gemm_bf16_launch(g, stream);
        if (shapes_h) {
            hipEventRecord(h1, stream); hipEventSynchronize(h1);
            float ms = 0.f; hipEventElapsedTime(&ms, h0, h1);
            shape_log(g, true, ms);
            hipEventDestroy(h0); hipEventDestroy(h1);
        }
        return rc;
    }
    auto al16 = [](const void* p, long stride_floats) { return ((uintptr_t)p & 15) == 0 && (stride_floats % 4) == 0; };
    const bool fast = g->K % BK == 0 && g->K <= ZERO_ROW_LEN && g->lda % 4 == 0 && g->ldw % 4 == 0 && al16(g->A, g->sA) && al16(g->W, g->sW) &&
                      (!g->A2 || al16(g->A2, g->sA));
    if (g->A_cat && (!fast || g->batch != 1 || g->seg_off || g->k_cat % BK || g->k_cat <= 0 || g->k_cat >= g->K || g->lda_cat % 4 ||
                     ((uintptr_t)g->A_cat & 15))) {
        roitr_set_error("roitr_gemm: K-concatenated A needs the fast path (K % 32, 16-byte rows), k_cat % 32 == 0, no batching", __FILE__, __LINE__);
        return ROITR_ERR_UNSUPPORTED;
    }
    if (g->a_cat_idx && !g->A_cat) { roitr_set_error("roitr_gemm: a_cat_idx without A_cat", __FILE__, __LINE__); return ROITR_ERR_ARG; }
    const int tn = g->ln_gamma ? g->N / BN : 1;   // LayerNorm epilogue: one block spans the row
    if (g->ln_gamma && (g->N % BN || (tn != 1 && tn != 2 && tn != 4) || g->batch != 1 || g->seg_off || g->relu || !g->ln_beta)) return ROITR_ERR_UNSUPPORTED;
    if (g->ip_feat && (!g->ln_gamma || !g->ip_idx || !g->ip_dist2)) { roitr_set_error("roitr_gemm: the interpolation addend rides in the fused LayerNorm epilogue only", __FILE__, __LINE__); return ROITR_ERR_UNSUPPORTED; }
    const int nx = div_up(g->N, BN * tn), ny = div_up(g->M, BM);
    const long Tl = (long)nx * ny * g->batch;
    if (Tl > 0x7ffffff0L) return ROITR_ERR_UNSUPPORTED;
    const int T = (int)Tl;
    const unsigned grid = (unsigned)xcd_grid(T);
    const int prof_cls = roitr_prof_is_enabled() ? roitr_gemm_prof_class(g) : ROITR_PROF_GEMM;
    if (g->batch_live)   // priced on the LIVE batches (device-side count), not on the capacity of the list
        roitr_prof_begin_live(prof_cls, 2.0 * g->M * g->N * (double)g->K, roitr_gemm_algorithmic_bytes(g) / g->batch, g->batch_live, stream);
    else roitr_prof_begin2(prof_cls, 2.0 * g->M * g->N * (double)g->K * g->batch, roitr_gemm_algorithmic_bytes(g), stream);
    // Measured and dropped (A/B on the forward bench): 64x128 / 128x128 multi-accumulator tiles (19.7 / 22.0 vs 16.9 ms of
    // GEMM per 128-pair forward), two K-slabs per barrier pair (7.9 vs 7.5 ms at 32 pairs), and a persistent-block
    // variant that opens the next tile (row pointers + first slab in flight) before the store epilogue (18.3-19.8 vs
    // 16.9 ms): at K = 64..512 the hardware dispatcher overlapping 7 resident 64x64 blocks per CU beats all of them.
    // Prefetch distance 2 (two alternating register sets) changes nothing either (45.7 vs 45.1 ms per forward): the
    // staging loads are not what the waves wait for.  Nor does distance 3 on the small grids of the one-pair-per-call mode
    // (round 2: 4.68 vs 4.65 ms per pair, bitwise the same results): those launches are not waiting for weight loads either.
    // Round 2, for the short-K layers of levels 1-2 (K = 64 / 128, 1.3 - 5.1 M rows; 1.8 - 3.3 TB/s, 40 - 80 TFLOP/s): a
    // weights-stationary persistent kernel -- the (64 TN x K) weight block staged once per workgroup, row tiles walked with the
    // next tile's rows prefetched across the MFMA block, two barriers per tile instead of two per slab, bitwise the results of
    // this kernel -- measured per shape at 512 pairs: N = 192 / 256 at K = 64 1.57 / 1.96 vs 1.61 / 2.10 ms, N = 384 / 512 at
    // K = 128 1.44 / 1.83 vs 1.30 / 1.68 ms, and the LayerNorm layers (N = 64 / 128, 1 - 2 resident workgroups per CU under
    // 57 - 156 KB of LDS) 1.8 / 1.4 vs 1.07 / 0.72 ms; gemm time per step 65.7 vs 55.2 ms.  Removed: re-staging the weights is
    // not what these launches wait for either.
    // The kernel alone reaches 106 TFLOP/s at K = 2048 and 78 at K = 256 (scripts/bench_gemm.py); the no-memory MFMA
    // ceiling measured on this part is 143-157 TFLOP/s (scripts/micro/mfma_peak.hip).
    // Also measured and removed (round 1-2): an LDS-DMA variant (global_load_lds_dwordx4 into a swizzled row-major image, two
    // stages, one barrier per slab): 98 vs 95 TFLOP/s at K = 2048 and 84 vs 80 at K = 256, but 62 vs 74 at K = 128 and 44 vs
    // 56 at K = 64 (32 KB of LDS per block: 5 instead of 7 resident blocks) -> 19.8 vs 17.9 ms of GEMM per 128-pair forward.
    // Round 3, measured and removed: s_setprio(1) around the 16 MFMAs of a slab (the waves in their MFMA phase first): 38.6 - 39.2 vs
    // 37.8 ms of GEMM per 512-pair step.
    // Round 3, measured and removed: a second kernel for K >= 128 / N >= 192 built like the on-chip GEMMs of local_block.hip
    // (v_mfma_f32_16x16x4_f32, weight fragments as float4 straight from L1 / L2, only A staged: 64 x 64-k slabs, double buffered,
    // one barrier per 64 k).  Correct (float64 test, all shapes of the forward) but 1.35-1.5x SLOWER on every shape it took
    // (M 319488 N 768 K 256: 1.67 vs 1.21 ms; M 1.28 M N 256 K 128: 1.31 vs 0.87 ms; gemm family 48.7 vs 37.6 ms per 512-pair
    // step): a lane's weight float4 comes from its own weight row, so one fragment load touches 16 cache lines at 64 B each --
    // the vector-memory pipe, not LDS, becomes the operand bottleneck once nothing else (attention, LayerNorm) overlaps it.
    static const bool shapes = getenv("ROITR_GEMM_SHAPES") != nullptr;  // debug: per-shape timing table at exit (synchronous)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (shapes) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, stream); }
    const bool a2 = g->A2 != nullptr;
    const bool il = T >= GEMM_IL_MIN_TILES;   // staging loads spread over the MFMA stream (see the kernel): large grids only
    if (fast && !g->ln_gamma && T < GEMM_SMALL_MAX_TILES) {   // a fraction of a tile per SIMD: 32 x 32 tiles with 4x shorter accumulator chains
        const int sx = div_up(g->N, SM), sy = div_up(g->M, SM);
        const int ST = sx * sy * g->batch;
        if (a2) gemm_small_kernel<true><<<xcd_grid(ST), 256, 0, stream>>>(*g, sx, sy, ST);
        else gemm_small_kernel<false><<<xcd_grid(ST), 256, 0, stream>>>(*g, sx, sy, ST);
    } else
#define GEMM_LAUNCH(F, TN_, LN_) \
    do { \
        if (a2 && il) gemm_kernel<F, TN_, LN_, true, true><<<grid, 256, 0, stream>>>(*g, nx, ny, T); \
        else if (a2) gemm_kernel<F, TN_, LN_, true, false><<<grid, 256, 0, stream>>>(*g, nx, ny, T); \
        else if (il) gemm_kernel<F, TN_, LN_, false, true><<<grid, 256, 0, stream>>>(*g, nx, ny, T); \
        else gemm_kernel<F, TN_, LN_, false, false><<<grid, 256, 0, stream>>>(*g, nx, ny, T); \
    } while (0)
    if (g->ln_gamma) {
        if (!fast) { if (tn != 1) return ROITR_ERR_UNSUPPORTED; gemm_kernel<false, 1, true><<<grid, 256, 0, stream>>>(*g, nx, ny, T); }
        else if (tn == 1) GEMM_LAUNCH(true, 1, true);
        else if (tn == 2) GEMM_LAUNCH(true, 2, true);
        else GEMM_LAUNCH(true, 4, true);
    } else if (fast) GEMM_LAUNCH(true, 1, false);
    else gemm_kernel<false, 1, false><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
#undef GEMM_LAUNCH
    if (shapes) {
        hipEventRecord(e1, stream); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); hipEventDestroy(e0); hipEventDestroy(e1);
        shape_log(g, fast, ms);
    }
    roitr_prof_end(prof_cls, stream);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

