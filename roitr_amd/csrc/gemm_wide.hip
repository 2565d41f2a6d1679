// fp32 linear layers with K >= 128 and N >= 192 (the matrix-shaped GEMMs of levels 3-4, the global transformer and the
// k | v projections): C = act( alpha * (A (+ A2)) @ W^T + bias ),  A: (M, K) rows (optionally gathered, optionally the
// K-concatenation [A | A_cat]), W: (N, K).
//
// Round 3.  gemm.hip stages BOTH operands of every 32-k slab through LDS behind two barriers; its own probes
// say the operand load path, not the MFMA feed, holds it at ~0.6 of the fp32 peak.  The fused block kernel (local_block.hip) runs
// its on-chip GEMMs ~1.2x faster with a different operand scheme, which this kernel applies to a plain GEMM:
//   * v_mfma_f32_16x16x4_f32; a wave owns 32 rows x (32 TN) columns (2 x 2 TN tiles) of the 64 x (64 TN) block tile;
//   * the WEIGHT fragments never touch LDS: lane (i = l & 15, g = l >> 4) reads one float4 = k 16c+4g.. of its weight row per
//     16-k chunk straight from L1 / L2 (weights are small and resident) and feeds component s to the s-th MFMA of the chunk;
//   * only A is staged: 64 rows x 64 k per slab, row-major with pitch 68 (conflict-free ds_read_b128 fragments), double
//     buffered -- ONE barrier per 64 k instead of two per 32 k;
//   * the finished tile leaves through LDS as coalesced float4 stores.
// A row's result is the same k-ordered chain whatever the tile, TN or M: batch-invariant like gemm.hip (not bitwise equal to
// it: inside a 16-k chunk the MFMA steps take k in the order 4g + s; the LayerNorm-fusable widths N = 64 / 128 therefore stay on
// gemm.hip, whose fused and two-launch forms are bitwise twins).
#include "common.h"
#include "roitr_engine.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WBM = 64, WKS = 64, WAP = WKS + 4;

template <int TN>
__global__ __launch_bounds__(256, TN == 4 ? 2 : 3) void gemm_wide_kernel(RoitrGemm g, int nx, int ny, int T)
{
    constexpr int BNW = 64 * TN;             // block tile columns
    constexpr int CT = 2 * TN;               // 16-column tiles per wave
    __shared__ __attribute__((aligned(16))) float As[2][WBM * WAP];
    const int tile = xcd_block_id(T);
    if (tile >= T) return;
    const int by_ = tile / nx, bx_ = tile - by_ * nx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, gq = lane >> 4;
    const int m0 = by_ * WBM, n0 = bx_ * BNW;
    const int r0 = (wave >> 1) * 32, c0 = n0 + (wave & 1) * 32 * TN;
    // ---- A staging: thread -> (row, 16-byte column piece) of the 64 x 64 slab, 4 pieces per thread
    const float* arow[4]; const float* arow2[4]; const float* arowc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = tid + 256 * u, r = e >> 4;
        const int am = m0 + r;
        arow[u] = nullptr; arow2[u] = nullptr; arowc[u] = nullptr;
        if (am < g.M) {
            const int src = g.a_idx ? g.a_idx[am] : am;
            if (src >= 0 && (g.a_limit <= 0 || src < g.a_limit)) {
                arow[u] = g.A + (size_t)src * g.lda + 4 * (e & 15);
                if (g.A2) arow2[u] = g.A2 + (size_t)src * g.lda + 4 * (e & 15);
                if (g.A_cat) arowc[u] = g.A_cat + (size_t)src * g.lda_cat + 4 * (e & 15);   // columns k >= k_cat of the product
            }
        }
    }
    auto fetch_a = [&](int k0, float4 (&v)[4]) {
        const bool cat = g.A_cat && k0 >= g.k_cat;             // slab-uniform: k_cat % 64 == 0
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* ap = cat ? (arowc[u] ? arowc[u] - g.k_cat : nullptr) : arow[u];
            v[u] = ap ? *reinterpret_cast<const float4*>(ap + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (arow2[u]) {
                const float4 t = *reinterpret_cast<const float4*>(arow2[u] + k0);
                v[u].x += t.x; v[u].y += t.y; v[u].z += t.z; v[u].w += t.w;
            }
        }
    };
    auto stage_a = [&](float* buf, const float4 (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u;
            *reinterpret_cast<float4*>(buf + (e >> 4) * WAP + 4 * (e & 15)) = v[u];
        }
    };
    // ---- weight rows of this lane: columns c0 + 16 t + i, t = 0 .. CT-1 (rows past N read row N-1: never stored)
    const float* wr[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int col = c0 + 16 * t + i;
        wr[t] = g.W + (size_t)(col < g.N ? col : g.N - 1) * g.ldw + 4 * gq;
    }
    f32x4 acc[2][CT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 av[4];
    fetch_a(0, av);
    stage_a(As[0], av);
    __syncthreads();
    const int nslab = g.K / WKS;
    for (int s = 0; s < nslab; ++s) {
        const float* A = As[s & 1];
        if (s + 1 < nslab) fetch_a((s + 1) * WKS, av);     // in flight across this slab's MFMAs
        const int k0 = s * WKS;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 b[CT];
#pragma unroll
            for (int t = 0; t < CT; ++t) b[t] = *reinterpret_cast<const float4*>(wr[t] + k0 + 16 * c);
            const float4 a0 = *reinterpret_cast<const float4*>(A + (r0 + i) * WAP + 16 * c + 4 * gq);
            const float4 a1 = *reinterpret_cast<const float4*>(A + (r0 + 16 + i) * WAP + 16 * c + 4 * gq);
#define GW_STEP(S)                                                                                   \
            _Pragma("unroll") for (int t = 0; t < CT; ++t) {                                         \
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.S, b[t].S, acc[0][t], 0, 0, 0);   \
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.S, b[t].S, acc[1][t], 0, 0, 0);   \
            }
            GW_STEP(x) GW_STEP(y) GW_STEP(z) GW_STEP(w)
#undef GW_STEP
        }
        if (s + 1 < nslab) stage_a(As[(s + 1) & 1], av);   // the other buffer: its readers finished before the last barrier
        __syncthreads();
    }
    // ---- epilogue: 64 x 64 column blocks through LDS (As[0] / As[1] are free), coalesced float4 stores
    float* tile_ = As[0];                                   // [64][WAP]
#pragma unroll
    for (int cb = 0; cb < TN; ++cb) {
        // the wave whose columns fall into block cb parks them: block cb = columns n0 + 64 cb .. +63 = waves with wn = cb / (TN/2)..
        // a wave's 32 TN columns span TN / 2 blocks (TN >= 2) or half a block (TN = 1)
        const int wcol0 = (wave & 1) * 32 * TN;             // first column of this wave inside the block tile
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int colt = wcol0 + 16 * t;                // tile's first column inside the block tile
            if (colt / 64 != cb) continue;
            const int col = colt - 64 * cb + i;
            const int gcol = n0 + colt + i;
            const float bv = (g.bias && gcol < g.N) ? g.bias[gcol] : 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[m][t][r] * g.alpha + bv;
                    if (g.relu) x = fmaxf(x, 0.f);
                    tile_[(r0 + 16 * m + 4 * gq + r) * WAP + col] = x;
                }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u, r = e >> 4, c4 = 4 * (e & 15);
            const int row = m0 + r, gcol = n0 + 64 * cb + c4;
            if (row < g.M && gcol < g.N) *reinterpret_cast<float4*>(g.C + (size_t)row * g.ldc + gcol) = *reinterpret_cast<const float4*>(tile_ + r * WAP + c4);
        }
        __syncthreads();
    }
}

}  // namespace

// Shapes this kernel takes (chosen by shape and flags only, never by M: a row's result must not depend on the batch it is in).
bool roitr_gemm_wide_takes(const RoitrGemm* g)
{
    if (g->A_cat && (g->A2 || g->k_cat % 64 || g->k_cat <= 0 || g->k_cat >= g->K || g->lda_cat % 4 || ((uintptr_t)g->A_cat & 15))) return false;
    return g->bf16 == 0 && g->batch == 1 && !g->seg_off && !g->w_idx && !g->ln_gamma && g->K >= 128 && g->K % 64 == 0 &&
           g->N >= 192 && g->N % 64 == 0 && g->lda % 4 == 0 && g->ldw % 4 == 0 && g->ldc % 4 == 0 &&
           (((uintptr_t)g->A | (uintptr_t)g->W | (uintptr_t)g->C | (uintptr_t)g->A2) & 15) == 0;
}

int roitr_gemm_wide_launch(const RoitrGemm* g, hipStream_t stream)
{
    // widest tile that still fills the chip (rows tiles x column tiles >= ~3 blocks per CU); narrower tiles for small M
    const int ny = div_up(g->M, WBM);
    int tn = 1;
    if (g->N % 256 == 0 && (long)ny * (g->N / 256) >= 768) tn = 4;
    else if (g->N % 128 == 0 && (long)ny * (g->N / 128) >= 768) tn = 2;
    const int nx = g->N / (64 * tn);
    const long Tl = (long)nx * ny;
    if (Tl > 0x7ffffff0L) return ROITR_ERR_UNSUPPORTED;
    const int T = (int)Tl;
    const unsigned grid = (unsigned)xcd_grid(T);
    if (tn == 4) gemm_wide_kernel<4><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
    else if (tn == 2) gemm_wide_kernel<2><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
    else gemm_wide_kernel<1><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
    return ROITR_OK;
}
