// bf16-operand linear layers on the gfx950 matrix cores (BASELINE.json configs[3]: "4DMatch ..., bf16").
//
// C[b] = act( alpha * (A[b] (+ A2[b])) @ W[b]^T + bias[b] )      A: (M,K)  W: (N,K) (torch Linear layout)
//
// Same contract as gemm.hip (RoitrGemm), selected by RoitrGemm::bf16 != 0: the products run on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate, 16x the fp32-MFMA rate), the weights are STORED in bf16 (converted once at
// roitr_engine_finalize), activations are either fp32 in HBM and rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while they are
// staged, or already stored in bf16 by the producing epilogue (ROITR_BF16_A / ROITR_BF16_C).  Bias, alpha, the LayerNorm
// epilogue and every reduction stay fp32.
//
// Tile 64 x 64 TN, 4 waves (one 32 x 32 TN strip each), BK = 64: per slab a wave issues 4 TN MFMAs of 32 cycles, so unlike
// the fp32 kernel (1024 MFMA cycles per 32-k slab) this one is bound by the operand path; the LDS image is plain row-major
// bf16 with a 144-byte row pitch: the 16-byte fragment reads of 16 consecutive rows fall on 16 distinct 4-bank groups
// (36 i mod 64 = 4 (9 i mod 16)), the 16-byte staging writes of a lane octet likewise.  A lane's MFMA operand is 8
// consecutive k (16 bytes): lanes 0..31 take k = 16 kk .. +7, lanes 32..63 k = 16 kk + 8 .. +7 of k-step kk.
#include "common.h"
#include "prof.h"
#include "roitr_engine.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 64, BN = 64, BK = 64;
constexpr int PITCH = 72;   // bf16 elements per LDS row (144 bytes)

__device__ __forceinline__ unsigned pack_bf16(float x, float y)   // low half = x; round to nearest even
{
    f32x2 v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ unsigned short to_bf16(float x) { return (unsigned short)(pack_bf16(x, 0.f) & 0xffffu); }

// 16 consecutive k of one row, as 8 packed bf16 pairs.  F32: the source is fp32 (four 16-byte loads, rounded here);
// otherwise bf16 (two 16-byte loads).  p == nullptr: a zero row.
struct Raw16 { uint4 a, b, c, d; };   // fp32: 16 floats; bf16: a, b only

template <bool F32>
__device__ __forceinline__ void load16(const void* p, long k, Raw16& r)
{
    if (p == nullptr) { r.a = r.b = r.c = r.d = make_uint4(0, 0, 0, 0); return; }
    if (F32) {
        const uint4* s = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(p) + k);
        r.a = s[0]; r.b = s[1]; r.c = s[2]; r.d = s[3];
    } else {
        const uint4* s = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p) + k);
        r.a = s[0]; r.b = s[1];
    }
}
__device__ __forceinline__ void add16(Raw16& r, const Raw16& o)   // fp32 payloads
{
    float* x = reinterpret_cast<float*>(&r);
    const float* y = reinterpret_cast<const float*>(&o);
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] += y[i];
}
template <bool F32>
__device__ __forceinline__ void store16(const Raw16& r, unsigned short* lds)   // lds: 16-byte aligned, 32 bytes written
{
    uint4* d = reinterpret_cast<uint4*>(lds);
    if (F32) {
        const float* x = reinterpret_cast<const float*>(&r);
        d[0] = make_uint4(pack_bf16(x[0], x[1]), pack_bf16(x[2], x[3]), pack_bf16(x[4], x[5]), pack_bf16(x[6], x[7]));
        d[1] = make_uint4(pack_bf16(x[8], x[9]), pack_bf16(x[10], x[11]), pack_bf16(x[12], x[13]), pack_bf16(x[14], x[15]));
    } else {
        d[0] = r.a; d[1] = r.b;
    }
}

// AF32: A (and A2) stored fp32 (else bf16); W is always stored bf16.  TN / LN as in gemm.hip.
template <bool AF32, int TN, bool LN>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(RoitrGemm g, int nx, int ny, int T)
{
    constexpr int TBN = BN * TN;
    constexpr int RP = TN == 4 ? 32 : 64;   // rows parked per LayerNorm pass
    constexpr int STAGE_BYTES = (BM + TBN) * PITCH * 2;
    constexpr int TILE_BYTES = LN ? RP * (TBN + 1) * 4 : (TN == 1 ? BM * (BN + 4) * 4 : 4 * 32 * 36 * 4);
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[STAGE_BYTES > TILE_BYTES ? STAGE_BYTES : TILE_BYTES];
    unsigned short* As = reinterpret_cast<unsigned short*>(smem_raw);
    unsigned short* Bs = As + BM * PITCH;
    if (g.batch_live) {   // batch list with a device-side live length: the tile map covers the live tiles only (all eight XCDs stay busy)
        const long tl = (long)*g.batch_live * nx * ny;
        if (tl < T) T = (int)tl;
    }
    const int tile = xcd_block_id_live(T);
    if (tile < 0) return;
    const int bz = tile / (nx * ny);
    const int rem = tile - bz * nx * ny;
    const int by_ = rem / nx, bx_ = rem - by_ * nx;
    const bool c_bf16 = (g.bf16 & ROITR_BF16_C) != 0;
    const size_t esz_a = AF32 ? 4 : 2;
    const char* A = reinterpret_cast<const char*>(g.A) + (size_t)bz * g.sA * esz_a;
    const char* A2 = g.A2 ? reinterpret_cast<const char*>(g.A2) + (size_t)bz * g.sA * esz_a : nullptr;
    const unsigned short* W = reinterpret_cast<const unsigned short*>(g.W) + (size_t)bz * g.sW;
    const float* bias = g.bias ? g.bias + (size_t)bz * g.sBias : nullptr;
    float* C = g.C + (size_t)bz * g.sC;                                                   // fp32 output
    unsigned short* Ch = reinterpret_cast<unsigned short*>(g.C) + (size_t)bz * g.sC;     // bf16 output
    const int* a_idx = g.a_idx ? g.a_idx + (size_t)bz * g.sAidx : nullptr;
    const int* w_idx = g.w_idx ? g.w_idx + (size_t)bz * g.sWidx : nullptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by_ * BM, n0 = bx_ * TBN;
    const int r = tid >> 2, kq = (tid & 3) * 16;   // staging: row r, 16 consecutive k from kq of the 64-k slab
    if (g.seg_off) {  // ragged batch: this batch's row segments of A and W
        const int ia = g.seg_a0 + bz, iw = g.seg_w0 + bz;
        const int a0 = ia == 0 ? 0 : g.seg_off[ia - 1], w0 = iw == 0 ? 0 : g.seg_off[iw - 1];
        g.M = g.seg_off[ia] - a0; g.N = g.seg_off[iw] - w0;
        A += (size_t)a0 * g.lda * esz_a; W += (size_t)w0 * g.ldw;
        if (A2) A2 += (size_t)a0 * g.lda * esz_a;
        if (m0 >= g.M || n0 >= g.N) return;  // block-uniform
    }

    const char* arow = nullptr; const char* arow2 = nullptr; const unsigned short* wrow[TN];
    {
        const int am = m0 + r;
        if (am < g.M) {
            const int src = a_idx ? a_idx[am] : am;
            if (src >= 0 && (g.a_limit <= 0 || src < g.a_limit)) {
                arow = A + (size_t)src * g.lda * esz_a;
                if (A2) arow2 = A2 + (size_t)src * g.lda * esz_a;
            }
        }
#pragma unroll
        for (int v = 0; v < TN; ++v) {
            wrow[v] = nullptr;
            const int wn_ = n0 + r + 64 * v;
            if (wn_ < g.N) {
                const int src = w_idx ? w_idx[wn_] : wn_;
                if (src >= 0 && (g.w_limit <= 0 || src < g.w_limit)) wrow[v] = W + (size_t)src * g.ldw;
            }
        }
    }

    f32x16 acc[TN];
#pragma unroll
    for (int v = 0; v < TN; ++v)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[v][i] = 0.f;

    Raw16 av, a2v, wv[TN];
    auto fetch = [&](int k) {
        load16<AF32>(arow, k, av);
        if (AF32 && A2) load16<true>(arow2, k, a2v);   // kernel-argument uniform
#pragma unroll
        for (int v = 0; v < TN; ++v) load16<false>(wrow[v], k, wv[v]);
    };
    fetch(kq);
    const int kh = lane >> 5, ml = lane & 31;
    const unsigned short* ar = As + (wm * 32 + ml) * PITCH + kh * 8;
    const unsigned short* br = Bs + (wn * 32 * TN + ml) * PITCH + kh * 8;
    unsigned short* aw = As + r * PITCH + kq;
    unsigned short* bw = Bs + r * PITCH + kq;
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        __syncthreads();
        if (AF32 && A2) add16(av, a2v);
        store16<AF32>(av, aw);
#pragma unroll
        for (int v = 0; v < TN; ++v) store16<false>(wv[v], bw + v * 64 * PITCH);
        __syncthreads();
        if (k0 + BK < g.K) fetch(k0 + BK + kq);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(ar + kk * 16);
            bf16x8 b[TN];
#pragma unroll
            for (int v = 0; v < TN; ++v) b[v] = *reinterpret_cast<const bf16x8*>(br + v * 32 * PITCH + kk * 16);
#pragma unroll
            for (int v = 0; v < TN; ++v) acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[v], acc[v], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // consumers of the prefetched registers stay below the MFMAs
    }
    if (LN) {
        float* tile_ = reinterpret_cast<float*>(smem_raw);   // [RP][TBN + 1]
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
                    if (c_bf16) {
                        // two columns per 4-byte store from the even lanes (round 6: 2-byte stores made this epilogue 2.6x slower than
                        // its fp32-output twin: 1.43 vs 0.55 ms per level-1 launch of the 64-pair 4DMatch call)
                        const float yo = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y), 0xB1, 0xf, 0xf, true));   // lane ^ 1
                        if ((g.ldc & 1) == 0) { if ((lane & 1) == 0) *reinterpret_cast<unsigned*>(Ch + (size_t)row * g.ldc + lane + 64 * i) = pack_bf16(y, yo); }
                        else Ch[(size_t)row * g.ldc + lane + 64 * i] = to_bf16(y);
                    } else C[(size_t)row * g.ldc + lane + 64 * i] = y;
                }
            }
        }
        return;
    }
    if (TN == 1 && n0 + BN <= g.N && (g.ldc & 3) == 0 && (((uintptr_t)g.C) & 15) == 0 && ((g.sC & 3) == 0)) {
        // full 64-column tile: transpose through the staging LDS, 16 (fp32) / 8 (bf16) bytes per lane and store
        constexpr int TP = BN + 4;
        __syncthreads();
        float* tile_ = reinterpret_cast<float*>(smem_raw);
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
            if (row < g.M) {
                const float4 x = *reinterpret_cast<const float4*>(tile_ + rl * TP + c4);
                if (c_bf16) *reinterpret_cast<uint2*>(Ch + (size_t)row * g.ldc + n0 + c4) = make_uint2(pack_bf16(x.x, x.y), pack_bf16(x.z, x.w));
                else *reinterpret_cast<float4*>(C + (size_t)row * g.ldc + n0 + c4) = x;
            }
        }
        return;
    }
    // TN > 1 (round 6: plain launches take 64 x 128 / 64 x 256 tiles where N allows -- the A rows are staged once per tile, and the
    // kernel is bound by its operand path): a wave's 32 x 32 blocks leave through its own 32 x 36 LDS scratch as 16-byte (fp32) /
    // 8-byte (bf16) stores, 8 lanes per row
    const bool wide = (g.ldc & 3) == 0 && (((uintptr_t)g.C) & 15) == 0 && ((g.sC & 3) == 0);
    if (TN > 1) __syncthreads();   // the operand images are dead
    float* sc = reinterpret_cast<float*>(smem_raw) + wave * 32 * 36;
#pragma unroll
    for (int v = 0; v < TN; ++v) {
        const int cb = n0 + (wn * TN + v) * 32;   // first column of this block
        if (cb >= g.N) continue;                  // wave-uniform
        const int col = cb + (lane & 31);
        const float bv = (bias && col < g.N) ? bias[col] : 0.f;
        if (TN > 1 && wide && cb + 32 <= g.N) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                float x = acc[v][i] * g.alpha + bv;
                if (g.relu) x = fmaxf(x, 0.f);
                sc[rl * 36 + (lane & 31)] = x;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int c4 = (lane & 7) * 4;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int rl = pass * 8 + (lane >> 3);
                const int row = m0 + wm * 32 + rl;
                if (row < g.M) {
                    const float4 x = *reinterpret_cast<const float4*>(sc + rl * 36 + c4);
                    if (c_bf16) *reinterpret_cast<uint2*>(Ch + (size_t)row * g.ldc + cb + c4) = make_uint2(pack_bf16(x.x, x.y), pack_bf16(x.z, x.w));
                    else *reinterpret_cast<float4*>(C + (size_t)row * g.ldc + cb + c4) = x;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the scratch is reused by the next column block
        } else if (col < g.N) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                if (row < g.M) {
                    float x = acc[v][i] * g.alpha + bv;
                    if (g.relu) x = fmaxf(x, 0.f);
                    if (c_bf16) Ch[(size_t)row * g.ldc + col] = to_bf16(x);
                    else C[(size_t)row * g.ldc + col] = x;
                }
            }
        }
    }
}

__global__ void f32_to_bf16_kernel(long n, const float* __restrict__ src, unsigned short* __restrict__ dst)
{
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < n) *reinterpret_cast<unsigned*>(dst + i) = pack_bf16(src[i], src[i + 1]);
    else if (i < n) dst[i] = to_bf16(src[i]);
}

}  // namespace

extern "C" int roitr_f32_to_bf16(long n, const float* src, unsigned short* dst, hipStream_t stream)
{
    if (n <= 0) return ROITR_OK;
    if (((uintptr_t)dst & 3) != 0) return ROITR_ERR_ARG;
    f32_to_bf16_kernel<<<(unsigned)((n / 2 + 1 + 255) / 256), 256, 0, stream>>>(n, src, dst);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

// shapes the bf16 kernel takes (the engine asks before it picks the bf16 weights of a layer)
extern "C" int roitr_gemm_bf16_supported(const RoitrGemm* g)
{
    const bool a_h = (g->bf16 & ROITR_BF16_A) != 0;
    const long a_al = a_h ? 8 : 4;   // elements per 16 bytes
    if (!(g->bf16 & ROITR_BF16_W)) return 0;                       // weights must be stored bf16
    if (g->A_cat) return 0;                                        // K-concatenated A: fp32 kernel only
    if (g->K <= 0 || g->K % BK) return 0;
    if (g->lda % a_al || g->ldw % 8 || g->sA % a_al || g->sW % 8) return 0;
    if (((uintptr_t)g->A & 15) || ((uintptr_t)g->W & 15) || (g->A2 && (((uintptr_t)g->A2 & 15) || a_h))) return 0;
    if (g->seg_off && (g->lda % a_al || g->ldw % 8)) return 0;
    if (g->ln_gamma) {
        const int tn = g->N / BN;
        if (g->N % BN || (tn != 1 && tn != 2 && tn != 4) || g->batch != 1 || g->seg_off || g->relu || !g->ln_beta) return 0;
    }
    return 1;
}

int roitr_gemm_bf16_launch(const RoitrGemm* g, hipStream_t stream)
{
    if (g->ip_feat) { roitr_set_error("roitr_gemm: the interpolation addend is an fp32-kernel epilogue", __FILE__, __LINE__); return ROITR_ERR_UNSUPPORTED; }
    if (!roitr_gemm_bf16_supported(g)) {
        roitr_set_error("roitr_gemm: shape / layout not supported by the bf16 kernel (K % 64, 16-byte rows, bf16 weights)", __FILE__, __LINE__);
        return ROITR_ERR_UNSUPPORTED;
    }
    // tile width: whole rows for the LayerNorm epilogue; plain launches 256 / 128 columns where N allows and the grid still fills the
    // chip (round 6: the kernel is bound by its operand path -- an A row is staged once per tile whatever its width; the result of an
    // element does not depend on the tile: same k order, same instruction)
    int tn = g->ln_gamma ? g->N / BN : 1;
    const int ny = div_up(g->M, BM);
    if (!g->ln_gamma && !g->seg_off && !g->w_idx) {
        tn = g->N % 256 == 0 ? 4 : (g->N % 128 == 0 ? 2 : 1);
        while (tn > 1 && (long)ny * div_up(g->N, BN * tn) * g->batch < 2048) tn >>= 1;
    }
    const int nx = div_up(g->N, BN * tn);
    const long Tl = (long)nx * ny * g->batch;
    if (Tl > 0x7ffffff0L) return ROITR_ERR_UNSUPPORTED;
    const int T = (int)Tl;
    const unsigned grid = (unsigned)xcd_grid(T);
    const bool a_h = (g->bf16 & ROITR_BF16_A) != 0;
    const int prof_cls = roitr_prof_is_enabled() ? roitr_gemm_prof_class(g) : ROITR_PROF_GEMM;
    if (g->batch_live)   // priced on the LIVE batches (device-side count), not on the capacity of the list
        roitr_prof_begin_live(prof_cls, 2.0 * g->M * g->N * (double)g->K, roitr_gemm_algorithmic_bytes(g) / g->batch, g->batch_live, stream);
    else roitr_prof_begin2(prof_cls, 2.0 * g->M * g->N * (double)g->K * g->batch, roitr_gemm_algorithmic_bytes(g), stream);
    if (g->ln_gamma) {
        if (a_h) {
            if (tn == 1) gemm_bf16_kernel<false, 1, true><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
            else if (tn == 2) gemm_bf16_kernel<false, 2, true><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
            else gemm_bf16_kernel<false, 4, true><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
        } else {
            if (tn == 1) gemm_bf16_kernel<true, 1, true><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
            else if (tn == 2) gemm_bf16_kernel<true, 2, true><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
            else gemm_bf16_kernel<true, 4, true><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
        }
    } else if (a_h) {
        if (tn == 1) gemm_bf16_kernel<false, 1, false><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
        else if (tn == 2) gemm_bf16_kernel<false, 2, false><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
        else gemm_bf16_kernel<false, 4, false><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
    } else {
        if (tn == 1) gemm_bf16_kernel<true, 1, false><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
        else if (tn == 2) gemm_bf16_kernel<true, 2, false><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
        else gemm_bf16_kernel<true, 4, false><<<grid, 256, 0, stream>>>(*g, nx, ny, T);
    }
    roitr_prof_end(prof_cls, stream);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}
