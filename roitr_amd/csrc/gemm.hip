// fp32 linear layers on the gfx950 matrix cores.
//
// C[b] = act( alpha * (A[b] (+ A2[b])) @ W[b]^T + bias[b] )      A: (M,K)  W: (N,K) (torch Linear layout)
//
// Every dense layer of the RoITr path goes through this one kernel (reference: the nn.Linear calls in
// model/transformer/*.py, model/model.py, model/RIGA_v2.py:64-68, and the einsum of RIGA_v2.py:150).
// v_mfma_f32_32x32x2_f32: exact fp32 FMA chains in k order (bit-reproducible, no TF32-style
// truncation exists on gfx950), 64 FLOP/clk/SIMD.  64x64 block tile, 4 waves (one 32x32 MFMA tile
// each), BK = 32, operands staged K-major in LDS with a +1 pad so that both the staging writes and
// the per-lane MFMA operand reads are bank-conflict free; the next K-slab is prefetched into
// registers while the current one feeds the MFMAs.
//
// Row gathers on either operand (with "index >= limit -> zero row", which is how the reference's
// padded patches are built, RIGA_v2.py:129-142) and an elementwise addend on A (x + pos of the
// cross-attention, geoattention.py:44-45) are fused into the staging loads.
#include "common.h"
#include "roitr_engine.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 32, LDP = 65;

struct RowSrc {
    const float* p;  // row pointer or null (zero row)
    const float* p2;
};

__device__ __forceinline__ void load8(const float* __restrict__ p, const float* __restrict__ p2, int k, int K, bool vec_ok, float (&v)[8])
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
        if (p2) {
            const float4 c = *reinterpret_cast<const float4*>(p2 + k);
            const float4 d = *reinterpret_cast<const float4*>(p2 + k + 4);
            v[0] += c.x; v[1] += c.y; v[2] += c.z; v[3] += c.w; v[4] += d.x; v[5] += d.y; v[6] += d.z; v[7] += d.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float x = 0.f;
            if (k + i < K) { x = p[k + i]; if (p2) x += p2[k + i]; }
            v[i] = x;
        }
    }
}

__global__ __launch_bounds__(256) void gemm_kernel(RoitrGemm g)
{
    __shared__ float As[BK][LDP];
    __shared__ float Bs[BK][LDP];
    const int bz = blockIdx.z;
    const float* A = g.A + (size_t)bz * g.sA;
    const float* A2 = g.A2 ? g.A2 + (size_t)bz * g.sA : nullptr;
    const float* W = g.W + (size_t)bz * g.sW;
    const float* bias = g.bias ? g.bias + (size_t)bz * g.sBias : nullptr;
    float* C = g.C + (size_t)bz * g.sC;
    const int* a_idx = g.a_idx ? g.a_idx + (size_t)bz * g.sAidx : nullptr;
    const int* w_idx = g.w_idx ? g.w_idx + (size_t)bz * g.sWidx : nullptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int r = tid >> 2, kq = (tid & 3) * 8;
    if (g.seg_off) {  // ragged batch: this batch's row segments of A and W
        const int ia = g.seg_a0 + bz, iw = g.seg_w0 + bz;
        const int a0 = ia == 0 ? 0 : g.seg_off[ia - 1], w0 = iw == 0 ? 0 : g.seg_off[iw - 1];
        g.M = g.seg_off[ia] - a0; g.N = g.seg_off[iw] - w0;
        A += (size_t)a0 * g.lda; W += (size_t)w0 * g.ldw;
        if (A2) A2 += (size_t)a0 * g.lda;
        if (m0 >= g.M || n0 >= g.N) return;  // block-uniform
    }

    const float* arow = nullptr; const float* arow2 = nullptr; const float* wrow = nullptr;
    {
        const int am = m0 + r;
        if (am < g.M) {
            int src = a_idx ? a_idx[am] : am;
            if (src >= 0 && (g.a_limit <= 0 || src < g.a_limit)) {
                arow = A + (size_t)src * g.lda;
                if (A2) arow2 = A2 + (size_t)src * g.lda;
            }
        }
        const int wn_ = n0 + r;
        if (wn_ < g.N) {
            int src = w_idx ? w_idx[wn_] : wn_;
            if (src >= 0 && (g.w_limit <= 0 || src < g.w_limit)) wrow = W + (size_t)src * g.ldw;
        }
    }
    const bool a_vec = (g.lda % 4 == 0) && (((uintptr_t)A & 15) == 0) && (!A2 || ((uintptr_t)A2 & 15) == 0);
    const bool w_vec = (g.ldw % 4 == 0) && (((uintptr_t)W & 15) == 0);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    float av[8], wv[8];
    load8(arow, arow2, kq, g.K, a_vec, av);
    load8(wrow, nullptr, kq, g.K, w_vec, wv);
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) { As[kq + i][r] = av[i]; Bs[kq + i][r] = wv[i]; }
        __syncthreads();
        if (k0 + BK < g.K) {
            load8(arow, arow2, k0 + BK + kq, g.K, a_vec, av);
            load8(wrow, nullptr, k0 + BK + kq, g.K, w_vec, wv);
        }
        const int kh = lane >> 5, ml = lane & 31;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float a = As[kk * 2 + kh][wm * 32 + ml];
            const float b = Bs[kk * 2 + kh][wn * 32 + ml];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < g.N) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            if (row < g.M) {
                float v = acc[i] * g.alpha + bv;
                if (g.relu) v = fmaxf(v, 0.f);
                C[(size_t)row * g.ldc + col] = v;
            }
        }
    }
}

}  // namespace

extern "C" int roitr_gemm(const RoitrGemm* g, hipStream_t stream)
{
    if (g->M <= 0 || g->N <= 0 || g->batch <= 0) return ROITR_OK;
    if (g->K <= 0 || !g->A || !g->W || !g->C) return ROITR_ERR_ARG;
    dim3 grid(div_up(g->N, BN), div_up(g->M, BM), g->batch);
    gemm_kernel<<<grid, 256, 0, stream>>>(*g);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}
