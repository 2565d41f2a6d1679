// Fused GeometricStructureEmbedding.forward (reference model/transformer/positional_encoding.py:139-154):
//
//     E[r, :] = proj_d(sinusoid(d_idx[r])) + max_k proj_a(sinusoid(a_idx[r, k]))          r = (cloud, i, j) pair row
//
// as ONE kernel.  The sinusoidal embedding (l.38-62: [sin(v w_0), cos(v w_0), sin(v w_1), ...]) is generated straight
// into the MFMA A-operand LDS image by the staging threads -- it never exists in HBM -- and the four projections of a
// 64-row tile (three angles, then the distance) run back to back on the same accumulator registers, with the max / add
// folded into the epilogue.  Replaces 2 sinusoid launches + 2 GEMMs + 1 combine launch and 7 (rows x C) fp32
// intermediates of the unfused sequence.
//
// Tile: 64 rows x (128 NJ) columns, 4 waves side by side along N, each 64 x (32 NJ) = 2 x NJ accumulators of
// v_mfma_f32_32x32x2_f32 (exact fp32 fma chains in k order, like gemm_kernel).  NJ = 1 is the default (occupancy beats
// the halved sin/cos work of NJ = 2, see the launcher).  The sin/cos of the NEXT k-slab are evaluated in
// the same basic block as the current slab's MFMAs (branch-free Cody-Waite + minimax polynomials), so the VALU work
// hides under the matrix pipe.
#include "common.h"
#include "prof.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 64, BK = 32, LDR = 20;  // same [kh][row][kk] LDS image as gemm.hip

// sin and cos of x, branch-free.  Three-term Cody-Waite reduction by pi/2 (exact products through fma; good to
// |x| ~ 1e5, far beyond d/sigma_d or angle/sigma_a of any scene) and the classic single-precision minimax polynomials
// on [-pi/4, pi/4]; within ~1 ulp of libm's sinf / cosf.
__device__ __forceinline__ void sincos_cw(float x, float& sn, float& cs)
{
    const float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(-k, 1.57079637050628662109375f, x);
    r = fmaf(-k, -4.37113900018624283e-8f, r);
    r = fmaf(-k, -1.71512449944201e-15f, r);
    const float z = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    ps = fmaf(ps * z, r, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    pc = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    const int q = (int)k;
    const float s1 = (q & 1) ? pc : ps, c1 = (q & 1) ? ps : pc;
    sn = (q & 2) ? -s1 : s1;
    cs = ((q + 1) & 2) ? -c1 : c1;
}

template <int NJ>
__global__ __launch_bounds__(256) void geo_embed_kernel(long rows, int C, int angle_k, const float* __restrict__ d_idx,
                                                        const float* __restrict__ a_idx, const float* __restrict__ div_term,
                                                        const float* __restrict__ Wd, const float* __restrict__ bd,
                                                        const float* __restrict__ Wa, const float* __restrict__ ba, float* __restrict__ out)
{
    constexpr int BNW = 128 * NJ;     // block tile width
    constexpr int NB = 2 * NJ;        // weight rows staged per thread
    __shared__ __attribute__((aligned(16))) float As[2 * BM * LDR];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BNW * LDR];
    __shared__ float divs[512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long m0 = (long)blockIdx.y * BM;
    const int n0 = blockIdx.x * BNW;
    const int r = tid >> 2, kq = (tid & 3) * 8;
    for (int i = tid; i < C / 2; i += 256) divs[i] = div_term[i];
    const long arow = m0 + r;
    const bool arow_ok = arow < rows;
    const int kh = lane >> 5, ml = lane & 31;
    const float4* ar = reinterpret_cast<const float4*>(As + (kh * BM + ml) * LDR);
    const float4* br = reinterpret_cast<const float4*>(Bs + (kh * BNW + wave * 32 * NJ + ml) * LDR);
    float4* aw0 = reinterpret_cast<float4*>(As + (0 * BM + r) * LDR + (kq >> 1));
    float4* aw1 = reinterpret_cast<float4*>(As + (1 * BM + r) * LDR + (kq >> 1));
    float4* bw0 = reinterpret_cast<float4*>(Bs + (0 * BNW + r) * LDR + (kq >> 1));
    float4* bw1 = reinterpret_cast<float4*>(Bs + (1 * BNW + r) * LDR + (kq >> 1));
    f32x16 amax[2][NJ], acc[2][NJ];
    __syncthreads();
    // passes 0..angle_k-1: the angle embeddings (running max); last pass: the distance embedding
    for (int pass = 0; pass <= angle_k; ++pass) {
        const bool dist = pass == angle_k;
        const float* wbase = (dist ? Wd : Wa) + (size_t)(n0 + r) * C + kq;
        float val = 0.f;
        if (arow_ok) val = dist ? d_idx[arow] : a_idx[arow * angle_k + pass];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        float4 w0[NB], w1[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            w0[b] = *reinterpret_cast<const float4*>(wbase + (size_t)b * 64 * C);
            w1[b] = *reinterpret_cast<const float4*>(wbase + (size_t)b * 64 * C + 4);
        }
        float sn[4], cs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sincos_cw(val * divs[(kq >> 1) + i], sn[i], cs[i]);
        for (int k0 = 0; k0 < C; k0 += BK) {
            __syncthreads();
            *aw0 = make_float4(sn[0], sn[1], sn[2], sn[3]);   // even k: sin
            *aw1 = make_float4(cs[0], cs[1], cs[2], cs[3]);   // odd k: cos
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                bw0[b * 64 * LDR / 4] = make_float4(w0[b].x, w0[b].z, w1[b].x, w1[b].z);
                bw1[b * 64 * LDR / 4] = make_float4(w0[b].y, w0[b].w, w1[b].y, w1[b].w);
            }
            __syncthreads();
            const bool more = k0 + BK < C;
            if (more) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    w0[b] = *reinterpret_cast<const float4*>(wbase + (size_t)b * 64 * C + k0 + BK);
                    w1[b] = *reinterpret_cast<const float4*>(wbase + (size_t)b * 64 * C + k0 + BK + 4);
                }
            }
            float4 af[2][4], bf[NJ][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i][q] = ar[i * 32 * LDR / 4 + q];
#pragma unroll
                for (int j = 0; j < NJ; ++j) bf[j][q] = br[j * 32 * LDR / 4 + q];
            }
            // next slab's embedding values: independent VALU work the scheduler can sink between the MFMAs
            const int f0 = ((more ? k0 + BK : 0) + kq) >> 1;
            float om[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) om[i] = val * divs[f0 + i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q].x, bf[j][q].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q].y, bf[j][q].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q].z, bf[j][q].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q].w, bf[j][q].w, acc[i][j], 0, 0, 0);
                    }
                sincos_cw(om[q], sn[q], cs[q]);
            }
        }
        if (!dist) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) amax[i][j][e] = pass == 0 ? acc[i][j][e] : fmaxf(amax[i][j][e], acc[i][j][e]);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = n0 + wave * 32 * NJ + j * 32 + (lane & 31);
        const float bdv = bd[col], bav = ba[col];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long row = m0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                // (acc_d + bd) + (max_k acc_k + ba): the association of the unfused path (bias in each projection, then
                // max, then add); max_k(acc_k + ba) == max_k(acc_k) + ba exactly because rounding is monotone
                if (row < rows) out[(size_t)row * C + col] = (acc[i][j][e] + bdv) + (amax[i][j][e] + bav);
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// An OPERATOR only (ops.geo_embed(split=True), tested against float64 in tests/test_stages_gpu.py), never used by the engine: the same embedding on the bf16 matrix cores with fp32-level
// accuracy.  Every fp32 operand is split into three bf16 pieces (x = hi + mid + lo, 8 + 8 + 8 mantissa bits, exact),
// and a product keeps the six piece pairs down to 2^-24: hh, hm, mh, hl, lh, mm -- six v_mfma_f32_32x32x16_bf16
// (fp32 accumulate, bf16 products are exact in fp32) instead of eight v_mfma_f32_32x32x2_f32 per 16 k, at 16x the
// rate per instruction: 2.7x less matrix-pipe time for an error of the order of fp32 rounding itself (measured against
// the fp32 kernel in tests/test_stages_gpu.py).  The weights are split once at engine finalize.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf16_rne(float x)
{
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l)
{
    h = bf16_rne(x);
    const float r1 = x - __uint_as_float(h << 16);
    m = bf16_rne(r1);
    const float r2 = r1 - __uint_as_float(m << 16);
    l = bf16_rne(r2);
}

__global__ void split3_bf16_kernel(long n, const float* __restrict__ src, unsigned short* __restrict__ dst)
{
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n) return;
    unsigned h, m, l;
    split3(src[i], h, m, l);
    dst[i] = (unsigned short)h; dst[n + i] = (unsigned short)m; dst[2 * n + i] = (unsigned short)l;
}


typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (x, y) -> packed bf16 pairs of the three pieces (v_cvt_pk_bf16_f32, round to nearest even); low half = x
__device__ __forceinline__ void split3_pair(float x, float y, unsigned& h, unsigned& m, unsigned& l)
{
    f32x2 v = {x, y};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    f32x2 r1 = {x - __uint_as_float(h << 16), y - __uint_as_float(h & 0xFFFF0000u)};
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
    f32x2 r2 = {r1[0] - __uint_as_float(m << 16), r1[1] - __uint_as_float(m & 0xFFFF0000u)};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ lds_ptr_t to_lds(const void* p) { return (lds_ptr_t)(unsigned)(uintptr_t)p; }

// Tile 64 x 128, 4 waves side by side along N (2 accumulators each).  Two LDS stages; the weight planes go global -> LDS
// by DMA (global_load_lds_dwordx4: with the matrix time cut 2.7x the ds_write path of the fp32 kernel would be the
// bottleneck), the generated A planes by ds_write_b128.  LDS rows are 4 chunks of 8 bf16, unpadded (the DMA writes
// linearly); chunk c of row r lives at physical chunk c ^ ((r >> 2) & 3): conflict-free b128 fragment reads.
// NP = 3: the split form above.  NP = 1: plain bf16 operands (one plane: the embedding and the weights rounded to bf16 once,
// fp32 accumulate) -- the bf16 operand mode of the engine (BASELINE config 4), W3 then is the (C, C) bf16 weight.
// (A device-function template behind two plain kernels: hipcc 7.2 silently emits no host stub for this body as a __global__
// template.)
template <int NP, bool OH = false>
__device__ __forceinline__ void geo_embed_split_body(long rows, int C, int angle_k, const float* __restrict__ d_idx,
                                                     const float* __restrict__ a_idx, const float* __restrict__ div_term,
                                                     const unsigned short* __restrict__ Wd3, const float* __restrict__ bd,
                                                     const unsigned short* __restrict__ Wa3, const float* __restrict__ ba,
                                                     float* __restrict__ out)
{
    constexpr int BNW = 128;
    __shared__ __attribute__((aligned(1024))) uint4 Bs[2][NP][BNW * 4];   // [stage][plane][row][chunk]
    __shared__ __attribute__((aligned(16))) uint4 As[2][NP][BM * 4];
    __shared__ float divs[512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long m0 = (long)blockIdx.y * BM;
    const int n0 = blockIdx.x * BNW;
    const int r = tid >> 2, ck = tid & 3, kq = ck * 8;
    for (int i = tid; i < C / 2; i += 256) divs[i] = div_term[i];
    const long arow = m0 + r;
    const bool arow_ok = arow < rows;
    const int kg = lane >> 5, ml = lane & 31;
    const size_t plane = (size_t)C * C;
    const int a_wr = r * 4 + (ck ^ ((r >> 2) & 3));
    // DMA assignment: 8 NP one-KB pieces per slab (NP planes x 8 groups of 16 rows), 2 NP per wave; lane -> row 16 u + lane / 4,
    // physical chunk lane % 4
    constexpr int PPW = 2 * NP;
    int dma_p[PPW], dma_lds[PPW]; size_t dma_src[PPW];
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
        const int idx = wave * PPW + t, p = idx >> 3, u = idx & 7;
        const int row = 16 * u + (lane >> 2);
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        dma_p[t] = p; dma_lds[t] = p * (BNW * 4) + u * 64;
        dma_src[t] = (size_t)p * plane + (size_t)(n0 + row) * C + 8 * lc;
    }
    f32x16 amax[2], acc[2];
    __syncthreads();
    for (int pass = 0; pass <= angle_k; ++pass) {
        const bool dist = pass == angle_k;
        const unsigned short* W3 = dist ? Wd3 : Wa3;
        float val = 0.f;
        if (arow_ok) val = dist ? d_idx[arow] : a_idx[arow * angle_k + pass];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        auto dma = [&](int stage, int k0) {
#pragma unroll
            for (int t = 0; t < PPW; ++t)
                __builtin_amdgcn_global_load_lds(W3 + dma_src[t] + k0, to_lds(&Bs[stage][0][0] + dma_lds[t]), 16, 0, 0);
        };
        auto gen = [&](int stage, int k0) {   // this thread's 8 consecutive k of the embedding row -> three bf16 planes
            const int f0 = (k0 + kq) >> 1;
            unsigned hh[4], mm[4], ll[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float sn, cs;
                sincos_cw(val * divs[f0 + i], sn, cs);
                if constexpr (NP == 3) split3_pair(sn, cs, hh[i], mm[i], ll[i]);
                else { f32x2 v = {sn, cs}; hh[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
            }
            As[stage][0][a_wr] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
            if constexpr (NP == 3) {
                As[stage][1][a_wr] = make_uint4(mm[0], mm[1], mm[2], mm[3]);
                As[stage][2][a_wr] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
            }
        };
        __syncthreads();   // the previous pass is done with both stages
        dma(0, 0);
        gen(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int stage = 0;
        for (int k0 = 0; k0 < C; k0 += BK) {
            const bool more = k0 + BK < C;
            if (more) dma(stage ^ 1, k0 + BK);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {   // two k-steps of 16 per slab
                const int ch = 2 * s2 + kg;
                bf16x8 a[NP][2], b[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    a[p][0] = __builtin_bit_cast(bf16x8, As[stage][p][ml * 4 + (ch ^ ((ml >> 2) & 3))]);
                    a[p][1] = __builtin_bit_cast(bf16x8, As[stage][p][(32 + ml) * 4 + (ch ^ ((ml >> 2) & 3))]);
                    const int br = wave * 32 + ml;
                    b[p] = __builtin_bit_cast(bf16x8, Bs[stage][p][br * 4 + (ch ^ ((br >> 2) & 3))]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {   // smallest terms first
                    if constexpr (NP == 3) {
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0], acc[i], 0, 0, 0);   // l h
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2], acc[i], 0, 0, 0);   // h l
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1], acc[i], 0, 0, 0);   // m m
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0], acc[i], 0, 0, 0);   // m h
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1], acc[i], 0, 0, 0);   // h m
                    }
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0], acc[i], 0, 0, 0);   // h h
                }
            }
            if (more) gen(stage ^ 1, k0 + BK);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            stage ^= 1;
        }
        if (!dist) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) amax[i][e] = pass == 0 ? acc[i][e] : fmaxf(amax[i][e], acc[i][e]);
        }
    }
    const int col = n0 + wave * 32 + (lane & 31);
    const float bdv = bd[col], bav = ba[col];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long row = m0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (row < rows) {
                const float val_ = (acc[i][e] + bdv) + (amax[i][e] + bav);
                if (OH) {   // E stored bf16 (round to nearest even)
                    f32x2 pr = {val_, 0.f};
                    reinterpret_cast<unsigned short*>(out)[(size_t)row * C + col] =
                        (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2)) & 0xffffu);
                } else out[(size_t)row * C + col] = val_;
            }
        }
}

__global__ __launch_bounds__(256) void geo_embed_split_kernel(long rows, int C, int angle_k, const float* __restrict__ d_idx,
                                                              const float* __restrict__ a_idx, const float* __restrict__ div_term,
                                                              const unsigned short* __restrict__ Wd3, const float* __restrict__ bd,
                                                              const unsigned short* __restrict__ Wa3, const float* __restrict__ ba,
                                                              float* __restrict__ out)
{
    geo_embed_split_body<3>(rows, C, angle_k, d_idx, a_idx, div_term, Wd3, bd, Wa3, ba, out);
}
__global__ __launch_bounds__(256) void geo_embed_bf16o_kernel(long rows, int C, int angle_k, const float* __restrict__ d_idx,
                                                              const float* __restrict__ a_idx, const float* __restrict__ div_term,
                                                              const unsigned short* __restrict__ Wd, const float* __restrict__ bd,
                                                              const unsigned short* __restrict__ Wa, const float* __restrict__ ba,
                                                              float* __restrict__ out)
{
    geo_embed_split_body<1, true>(rows, C, angle_k, d_idx, a_idx, div_term, Wd, bd, Wa, ba, out);
}
__global__ __launch_bounds__(256) void geo_embed_bf16_kernel(long rows, int C, int angle_k, const float* __restrict__ d_idx,
                                                             const float* __restrict__ a_idx, const float* __restrict__ div_term,
                                                             const unsigned short* __restrict__ Wd, const float* __restrict__ bd,
                                                             const unsigned short* __restrict__ Wa, const float* __restrict__ ba,
                                                             float* __restrict__ out)
{
    geo_embed_split_body<1>(rows, C, angle_k, d_idx, a_idx, div_term, Wd, bd, Wa, ba, out);
}

}  // namespace

// d_idx (rows), a_idx (rows, angle_k), div_term (C/2), proj_d / proj_a weights (C,C) + biases, out (rows, C)
extern "C" int roitr_geo_embed(long rows, int C, int angle_k, const float* d_idx, const float* a_idx, const float* div_term,
                               const float* Wd, const float* bd, const float* Wa, const float* ba, float* out, hipStream_t stream)
{
    if (rows <= 0) return ROITR_OK;
    if (C % 128 || C > 1024 || angle_k < 1) return ROITR_ERR_UNSUPPORTED;
    const long mt = (rows + BM - 1) / BM;
    if (mt > 0x7fffffffL) return ROITR_ERR_UNSUPPORTED;
    // measured (B=32, N=5000, C=256): NJ=1 1.90 ms (156 VGPRs, 3 blocks/CU), NJ=2 2.27 ms (284 VGPRs, 1 block/CU).
    // Also measured and dropped: one staged weight slab shared by the three angle passes (3 x 2 accumulators per wave, 199
    // VGPRs, 2 blocks/CU): 7.32 vs 7.37 ms per 128-pair forward -- weight traffic / barriers are not what holds the kernel
    // at 66 % of the MFMA peak; NJ=1 with 128 VGPRs forced (4 blocks/CU, 92 spills): 7.76 ms.
    roitr_prof_begin(ROITR_PROF_GEO_EMBED, 2.0 * rows * (1.0 + angle_k) * (double)C * C, stream);
    geo_embed_kernel<1><<<dim3(C / 128, (unsigned)mt), 256, 0, stream>>>(rows, C, angle_k, d_idx, a_idx, div_term, Wd, bd, Wa, ba, out);
    roitr_prof_end(ROITR_PROF_GEO_EMBED, stream);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}


/* OPT-IN split-bf16 form of roitr_geo_embed (see geo_embed_split_kernel): W*3 = the three bf16 planes of the (C, C) weight
 * made by roitr_split3_bf16 (3 * C * C uint16). */
extern "C" int roitr_split3_bf16(long n, const float* src, unsigned short* dst, hipStream_t stream)
{
    if (n <= 0) return ROITR_OK;
    split3_bf16_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(n, src, dst);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_geo_embed_split(long rows, int C, int angle_k, const float* d_idx, const float* a_idx, const float* div_term,
                                     const unsigned short* Wd3, const float* bd, const unsigned short* Wa3, const float* ba, float* out,
                                     hipStream_t stream)
{
    if (rows <= 0) return ROITR_OK;
    if (C % 128 || C > 1024 || angle_k < 1) return ROITR_ERR_UNSUPPORTED;
    const long mt = (rows + BM - 1) / BM;
    if (mt > 0x7fffffffL) return ROITR_ERR_UNSUPPORTED;
    roitr_prof_begin(ROITR_PROF_GEO_EMBED, 2.0 * rows * (1.0 + angle_k) * (double)C * C, stream);
    geo_embed_split_kernel<<<dim3(C / 128, (unsigned)mt), 256, 0, stream>>>(rows, C, angle_k, d_idx, a_idx, div_term, Wd3, bd, Wa3, ba, out);
    roitr_prof_end(ROITR_PROF_GEO_EMBED, stream);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

/* bf16 operand form of roitr_geo_embed (engine operand_dtype = bf16): Wd / Wa are the (C, C) weights stored in bf16
 * (roitr_f32_to_bf16), the generated sinusoid is rounded to bf16 as it is written into the MFMA image, fp32 accumulate,
 * fp32 max / add epilogue. */
extern "C" int roitr_geo_embed_bf16(long rows, int C, int angle_k, const float* d_idx, const float* a_idx, const float* div_term,
                                    const unsigned short* Wd, const float* bd, const unsigned short* Wa, const float* ba, float* out,
                                    hipStream_t stream)
{
    if (rows <= 0) return ROITR_OK;
    if (C % 128 || C > 1024 || angle_k < 1) return ROITR_ERR_UNSUPPORTED;
    const long mt = (rows + BM - 1) / BM;
    if (mt > 0x7fffffffL) return ROITR_ERR_UNSUPPORTED;
    roitr_prof_begin(ROITR_PROF_GEO_EMBED, 2.0 * rows * (1.0 + angle_k) * (double)C * C, stream);
    geo_embed_bf16_kernel<<<dim3(C / 128, (unsigned)mt), 256, 0, stream>>>(rows, C, angle_k, d_idx, a_idx, div_term, Wd, bd, Wa, ba, out);
    roitr_prof_end(ROITR_PROF_GEO_EMBED, stream);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_geo_embed_bf16_out(long rows, int C, int angle_k, const float* d_idx, const float* a_idx, const float* div_term,
                                        const unsigned short* Wd, const float* bd, const unsigned short* Wa, const float* ba,
                                        unsigned short* out, hipStream_t stream)
{
    if (rows <= 0) return ROITR_OK;
    if (C % 128 || C > 1024 || angle_k < 1) return ROITR_ERR_UNSUPPORTED;
    const long mt = (rows + BM - 1) / BM;
    if (mt > 0x7fffffffL) return ROITR_ERR_UNSUPPORTED;
    roitr_prof_begin(ROITR_PROF_GEO_EMBED, 2.0 * rows * (1.0 + angle_k) * (double)C * C, stream);
    geo_embed_bf16o_kernel<<<dim3(C / 128, (unsigned)mt), 256, 0, stream>>>(rows, C, angle_k, d_idx, a_idx, div_term, Wd, bd, Wa, ba,
                                                                           reinterpret_cast<float*>(out));
    roitr_prof_end(ROITR_PROF_GEO_EMBED, stream);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}
