// Standalone probe for the round-4 fp32 GEMM generation (no torch, no Python): candidate kernels are timed per shape and
// compared BITWISE with libroitr_hip.so's roitr_gemm (same k order => interchangeable per row, whatever the row count).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I roitr_amd/csrc -o scripts/micro/gemm_gen2 scripts/micro/gemm_gen2.hip -ldl
//   scripts/micro/gemm_gen2 [roitr_amd/lib/libroitr_hip.so]
// Variants:
//   R<TM,TN>: register-staged (global -> VGPR -> ds_write_b128 into the [kh][row][kk] image of gemm.hip), every wave owns
//             TM x TN accumulators of 32x32 (block tile 64 TM x 64 TN), two barriers per 32-k slab;
//   D<TM,TN>: LDS-DMA staged (global_load_lds_dwordx4 into an XOR-swizzled row-major image, two stages, ONE barrier per slab),
//             a lane reads 16-byte pieces and picks its k-parity element with v_cndmask -- the k order of the register kernel.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "roitr_engine.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDR = 20;
__device__ __attribute__((aligned(16))) float g_zero_row[4096];

__host__ __device__ inline int xcd_grid(int n) { return ((n + 7) >> 3) << 3; }
__device__ inline int xcd_block_id(int n) { const int b = blockIdx.x; return (b & 7) * ((n + 7) >> 3) + (b >> 3); }

struct P { const float* A; const float* W; const float* bias; float* C; int M, N, K, lda, ldw, ldc, relu; };

// epilogue shared by both variants: a wave's TM x TN accumulators -> bias / relu -> per-wave LDS scratch -> 16-byte stores
template <int TM, int TN>
__device__ __forceinline__ void store_tiles(const P& g, f32x16 (&acc)[TM][TN], float* scratch, int m0, int n0, int wm, int wn, int lane)
{
    constexpr int PITCH = 32 * TN + 4;
    float* sc = scratch;   // this wave's 32 x PITCH floats
#pragma unroll
    for (int t = 0; t < TM; ++t) {
#pragma unroll
        for (int v = 0; v < TN; ++v) {
            const int col = n0 + (wn * TN + v) * 32 + (lane & 31);
            const float bv = (g.bias && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                float x = acc[t][v][i] + bv;
                if (g.relu) x = fmaxf(x, 0.f);
                sc[rl * PITCH + v * 32 + (lane & 31)] = x;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int L4 = 8 * TN;            // float4 per row
        constexpr int RPP = 64 / L4;          // rows per pass
        const int c4 = (lane % L4) * 4;
#pragma unroll
        for (int pass = 0; pass < 32 / RPP; ++pass) {
            const int rl = pass * RPP + lane / L4;
            const int row = m0 + (wm * TM + t) * 32 + rl;
            const int col = n0 + wn * 32 * TN + c4;
            if (row < g.M && col + 3 < g.N)
                *reinterpret_cast<float4*>(g.C + (size_t)row * g.ldc + col) = *reinterpret_cast<const float4*>(sc + rl * PITCH + c4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ---------------------------------------------------------------------------------------------- register-staged
// AB (ablation, timing only -- results are wrong for AB > 0): 1 = no global loads inside the loop, 2 = also no LDS writes / barriers,
// 3 = also no fragment reads (pure MFMA + epilogue), 4 = also no epilogue stores
template <int TM, int TN, int AB = 0>
__global__ __launch_bounds__(256) void gemm_r_kernel(P g, int nx, int T)
{
    constexpr int BMt = 64 * TM, BNt = 64 * TN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BMt * LDR;
    const int tile = xcd_block_id(T);
    if (tile >= T) return;
    const int by_ = tile / nx, bx_ = tile - by_ * nx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by_ * BMt, n0 = bx_ * BNt;
    const int r = tid >> 2, j = tid & 3;
    const float* arow[TM]; const float* wrow[TN];
#pragma unroll
    for (int u = 0; u < TM; ++u) { const int am = m0 + r + 64 * u; arow[u] = (am < g.M ? g.A + (size_t)am * g.lda : g_zero_row) + 4 * j; }
#pragma unroll
    for (int v = 0; v < TN; ++v) { const int wn_ = n0 + r + 64 * v; wrow[v] = (wn_ < g.N ? g.W + (size_t)wn_ * g.ldw : g_zero_row) + 4 * j; }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int v = 0; v < TN; ++v)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][v][i] = 0.f;
    float4 ax[TM], ay[TM], wx[TN], wy[TN];
    auto fetch = [&](int k) {
        const int ka = (AB == 5 || AB == 6) ? 0 : k, kw = (AB == 5 || AB == 7) ? 0 : k;   // 5: both operands cache-hot, 6: A hot, 7: W hot
#pragma unroll
        for (int u = 0; u < TM; ++u) { ax[u] = *reinterpret_cast<const float4*>(arow[u] + ka); ay[u] = *reinterpret_cast<const float4*>(arow[u] + ka + 16); }
#pragma unroll
        for (int v = 0; v < TN; ++v) { wx[v] = *reinterpret_cast<const float4*>(wrow[v] + kw); wy[v] = *reinterpret_cast<const float4*>(wrow[v] + kw + 16); }
    };
    fetch(0);
    const int kh = lane >> 5, ml = lane & 31;
    const float4* ar = reinterpret_cast<const float4*>(As + (kh * BMt + wm * 32 * TM + ml) * LDR);
    const float4* br = reinterpret_cast<const float4*>(Bs + (kh * BNt + wn * 32 * TN + ml) * LDR);
    float4* aw0 = reinterpret_cast<float4*>(As + (0 * BMt + r) * LDR + 4 * j);
    float4* aw1 = reinterpret_cast<float4*>(As + (1 * BMt + r) * LDR + 4 * j);
    float4* bw0 = reinterpret_cast<float4*>(Bs + (0 * BNt + r) * LDR + 4 * j);
    float4* bw1 = reinterpret_cast<float4*>(Bs + (1 * BNt + r) * LDR + 4 * j);
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f dx[TM], dy[TM], ex[TN], ey[TN];
    float4 caf[TM], cbf[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) caf[t] = ax[t];
#pragma unroll
    for (int v = 0; v < TN; ++v) cbf[v] = wx[v];
    for (int k0 = 0; k0 < g.K; k0 += 32) {
        if (AB < 2 || AB >= 5 || k0 == 0) {   // (AB 8 writes the stale prologue registers like AB 1)
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TM; ++u) {
            aw0[u * 64 * LDR / 4] = make_float4(ax[u].x, ax[u].z, ay[u].x, ay[u].z);
            aw1[u * 64 * LDR / 4] = make_float4(ax[u].y, ax[u].w, ay[u].y, ay[u].w);
        }
#pragma unroll
        for (int v = 0; v < TN; ++v) {
            bw0[v * 64 * LDR / 4] = make_float4(wx[v].x, wx[v].z, wy[v].x, wy[v].z);
            bw1[v * 64 * LDR / 4] = make_float4(wx[v].y, wx[v].w, wy[v].y, wy[v].w);
        }
        __syncthreads();
        }
        if ((AB < 1 || (AB >= 5 && AB <= 7)) && k0 + 32 < g.K) fetch(k0 + 32);
        if (AB == 8) {   // loads ISSUED (inline asm: the compiler neither tracks nor waits for them) but never consumed inside the loop
#pragma unroll
            for (int u = 0; u < TM; ++u) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dx[u]) : "v"(arow[u] + k0) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(dy[u]) : "v"(arow[u] + k0) : "memory");
            }
#pragma unroll
            for (int v = 0; v < TN; ++v) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ex[v]) : "v"(wrow[v] + k0) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(ey[v]) : "v"(wrow[v] + k0) : "memory");
            }
        }
        float4 afq[4][TM], bfq[4][TN];
        if (AB == 9) {   // every fragment of the slab up front (like the TN = 1 path of gemm.hip): nothing but MFMAs and the
                         // staging loads below
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int t = 0; t < TM; ++t) afq[q][t] = ar[t * 32 * LDR / 4 + q];
#pragma unroll
                for (int v = 0; v < TN; ++v) bfq[q][v] = br[v * 32 * LDR / 4 + q];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) af[t] = AB == 9 ? afq[q][t] : (AB == 3 || AB == 4) ? caf[t] : ar[t * 32 * LDR / 4 + q];
#pragma unroll
            for (int v = 0; v < TN; ++v) bf[v] = AB == 9 ? bfq[q][v] : (AB == 3 || AB == 4) ? cbf[v] : br[v * 32 * LDR / 4 + q];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int v = 0; v < TN; ++v) {
                        const float a = c == 0 ? af[t].x : c == 1 ? af[t].y : c == 2 ? af[t].z : af[t].w;
                        const float b = c == 0 ? bf[v].x : c == 1 ? bf[v].y : c == 2 ? bf[v].z : bf[v].w;
                        acc[t][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t][v], 0, 0, 0);
                        if (AB == 9) {   // the slab's 2 (TM + TN) staging loads spread over its MFMAs, one at a time
                            constexpr int TOT = 16 * TM * TN, NS = 2 * (TM + TN);
                            const int n = ((q * 4 + c) * TM + t) * TN + v;
#pragma unroll
                            for (int i = 0; i < NS; ++i)
                                if (n == (i * TOT + TOT / 2) / NS) {
                                    __builtin_amdgcn_sched_barrier(0);
                                    const int kn = k0 + 32 < g.K ? k0 + 32 : k0;
                                    if (i < 2 * TM) { if (i & 1) ay[i >> 1] = *reinterpret_cast<const float4*>(arow[i >> 1] + kn + 16); else ax[i >> 1] = *reinterpret_cast<const float4*>(arow[i >> 1] + kn); }
                                    else { const int w_ = i - 2 * TM; if (w_ & 1) wy[w_ >> 1] = *reinterpret_cast<const float4*>(wrow[w_ >> 1] + kn + 16); else wx[w_ >> 1] = *reinterpret_cast<const float4*>(wrow[w_ >> 1] + kn); }
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                        }
                    }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    store_tiles<TM, TN>(g, acc, smem + wave * 32 * (32 * TN + 4), m0, n0, wm, wn, lane);
}


// ---------------------------------------------------------------------------------------------- register-staged, software-pipelined
// Two LDS stages, ONE raw barrier per slab placed before the last quarter of the slab's MFMAs: a wave's MFMA stream never stops
// for staging -- slab s+1 is written to the other stage and slab s+2's global loads are issued at the top of slab s, the
// fragments of quarter q+1 are read while quarter q multiplies, and the first quarter of slab s+1 is read right after the
// barrier, under the last quarter of slab s.
// LDS image: row-major [row][32 k] with a 36-float pitch (16 rows x 16 B hit 64 distinct banks: conflict-free ds_read_b128
// for any 16 rows distinct mod 16), written by the loaded float4 AS IS (8 lanes = one row's 128 B: no register shuffle, full
// cache lines per row); a lane reads whole 16-byte pieces and takes its k-parity element with v_cndmask -- the k order of the
// register kernel of gemm.hip (piece order 0 4 1 5 2 6 3 7, inside a piece k, k+1 | k+2, k+3).
constexpr int SP = 36;
typedef float f4 __attribute__((ext_vector_type(4)));   // native vector: hipcc keeps it in registers (a float4 STRUCT copy between
                                                          // address spaces becomes a memcpy through scratch)
template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_s_kernel(P g, int nx, int T)
{
    constexpr int BMt = 64 * TM, BNt = 64 * TN;
    constexpr int STG = (BMt + BNt) * SP;          // floats per stage
    constexpr int NA = BMt / 32, NB = BNt / 32;    // f4 loads per thread and slab
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tile = xcd_block_id(T);
    if (tile >= T) return;
    const int by_ = tile / nx, bx_ = tile - by_ * nx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by_ * BMt, n0 = bx_ * BNt;
    const int r = tid >> 3, pc = tid & 7;          // staging: row r (+ 32 u), piece pc
    const float* arow[NA]; const float* wrow[NB];
#pragma unroll
    for (int u = 0; u < NA; ++u) { const int am = m0 + r + 32 * u; arow[u] = (am < g.M ? g.A + (size_t)am * g.lda : g_zero_row) + 4 * pc; }
#pragma unroll
    for (int u = 0; u < NB; ++u) { const int wn_ = n0 + r + 32 * u; wrow[u] = (wn_ < g.N ? g.W + (size_t)wn_ * g.ldw : g_zero_row) + 4 * pc; }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int v = 0; v < TN; ++v)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][v][i] = 0.f;
    f4 av[NA], wv[NB];
    auto fetch = [&](int k) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NA; ++u) av[u] = *reinterpret_cast<const f4*>(arow[u] + k);
#pragma unroll
        for (int u = 0; u < NB; ++u) wv[u] = *reinterpret_cast<const f4*>(wrow[u] + k);
    };
    const int kh = lane >> 5, ml = lane & 31;
    const int a_rd = (wm * 32 * TM + ml) * SP, b_rd = (BMt + wn * 32 * TN + ml) * SP;     // float offsets inside a stage
    const int a_wr = r * SP + 4 * pc, b_wr = (BMt + r) * SP + 4 * pc;
    auto stage_write = [&](float* st) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NA; ++u) *reinterpret_cast<f4*>(st + a_wr + 32 * u * SP) = av[u];
#pragma unroll
        for (int u = 0; u < NB; ++u) *reinterpret_cast<f4*>(st + b_wr + 32 * u * SP) = wv[u];
    };
    // quarter q of a slab = pieces q and q + 4
    auto frag_read = [&](const float* st, int q, f4 (&fa)[2][TM], f4 (&fb)[2][TN]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int t = 0; t < TM; ++t) fa[h][t] = *reinterpret_cast<const f4*>(st + a_rd + t * 32 * SP + 4 * (q + 4 * h));
#pragma unroll
            for (int v = 0; v < TN; ++v) fb[h][v] = *reinterpret_cast<const f4*>(st + b_rd + v * 32 * SP + 4 * (q + 4 * h));
        }
    };
    auto mma = [&](const f4 (&fa)[2][TM], const f4 (&fb)[2][TN]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float a[TM], b[TN];
#pragma unroll
                for (int t = 0; t < TM; ++t) a[t] = c == 0 ? (kh ? fa[h][t].y : fa[h][t].x) : (kh ? fa[h][t].w : fa[h][t].z);
#pragma unroll
                for (int v = 0; v < TN; ++v) b[v] = c == 0 ? (kh ? fb[h][v].y : fb[h][v].x) : (kh ? fb[h][v].w : fb[h][v].z);
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int v = 0; v < TN; ++v) acc[t][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[v], acc[t][v], 0, 0, 0);
            }
    };
    const int S = g.K / 32;
    fetch(0);
    stage_write(smem);
    fetch(S > 1 ? 32 : 0);
    __syncthreads();
    f4 fa0[2][TM], fb0[2][TN], fa1[2][TM], fb1[2][TN];
    frag_read(smem, 0, fa0, fb0);
    // No conditionals in the loop body (a conditional fetch / write makes the prefetch registers phi nodes: hipcc then copies the
    // loaded values right behind the loads and waits for them): the last iterations re-fetch the last slab and write a stage
    // nobody reads any more.
    for (int s = 0; s < S; ++s) {
        float* cur = smem + (s & 1) * STG;
        float* nxt = smem + ((s & 1) ^ 1) * STG;
        stage_write(nxt);
        fetch((s + 2 < S ? s + 2 : S - 1) * 32);
        __builtin_amdgcn_sched_barrier(0);
        frag_read(cur, 1, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        frag_read(cur, 2, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        frag_read(cur, 3, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        frag_read(nxt, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    store_tiles<TM, TN>(g, acc, smem + wave * 32 * (32 * TN + 4), m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------- LDS-DMA staged
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ lds_ptr_t to_lds(const void* p) { return (lds_ptr_t)(unsigned)(uintptr_t)p; }

template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_d_kernel(P g, int nx, int T)
{
#if defined(__HIP_DEVICE_COMPILE__)   // the LDS-DMA builtin exists in the device pass only
    constexpr int BMt = 64 * TM, BNt = 64 * TN;
    constexpr int STAGE = (BMt + BNt) * 32;          // floats per stage: row-major [row][32 k], A rows then W rows
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int tile = xcd_block_id(T);
    if (tile >= T) return;
    const int by_ = tile / nx, bx_ = tile - by_ * nx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by_ * BMt, n0 = bx_ * BNt;
    // staging: one wave instruction fills 8 rows x 128 B; wave w takes the row groups w, w + 4, ...; lane t -> row t / 8, slot t % 8,
    // the slot holds the logical piece slot ^ ((row >> 1) & 7)
    constexpr int NA = BMt / 32, NB = BNt / 32;      // instructions per wave and operand
    const float* asrc[NA]; const float* wsrc[NB];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int rr = 8 * (4 * u + wave) + (lane >> 3);
        const int lc = (lane & 7) ^ ((rr >> 1) & 7);
        const int am = m0 + rr;
        asrc[u] = (am < g.M ? g.A + (size_t)am * g.lda : g_zero_row) + 4 * lc;
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int rr = 8 * (4 * u + wave) + (lane >> 3);
        const int lc = (lane & 7) ^ ((rr >> 1) & 7);
        const int wr = n0 + rr;
        wsrc[u] = (wr < g.N ? g.W + (size_t)wr * g.ldw : g_zero_row) + 4 * lc;
    }
    auto issue = [&](int stage, int k0) {
        float* sa = smem + stage * STAGE;
        float* sb = sa + BMt * 32;
#pragma unroll
        for (int u = 0; u < NA; ++u) __builtin_amdgcn_global_load_lds(asrc[u] + k0, to_lds(sa + (4 * u + wave) * 256), 16, 0, 0);
#pragma unroll
        for (int u = 0; u < NB; ++u) __builtin_amdgcn_global_load_lds(wsrc[u] + k0, to_lds(sb + (4 * u + wave) * 256), 16, 0, 0);
    };
    const int kh = lane >> 5, ml = lane & 31;
    const int sw = (ml >> 1) & 7;          // row blocks start at multiples of 32 rows: the swizzle term is the lane's
    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int v = 0; v < TN; ++v)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][v][i] = 0.f;
    issue(0, 0);
    int stage = 0;
    for (int k0 = 0; k0 < g.K; k0 += 32) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of slab k0 have landed ...
        __syncthreads();                                      // ... everyone's have, and everyone is done reading the other stage
        if (k0 + 32 < g.K) issue(stage ^ 1, k0 + 32);
        const float* sa = smem + stage * STAGE + (wm * 32 * TM + ml) * 32;
        const float* sb = smem + stage * STAGE + BMt * 32 + (wn * 32 * TN + ml) * 32;
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) {
            const int q = (idx >> 1) + 4 * (idx & 1);    // piece order of the register kernel: 0 4 1 5 2 6 3 7
            float4 af[TM], bf[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) af[t] = *reinterpret_cast<const float4*>(sa + t * 32 * 32 + 4 * (q ^ sw));
#pragma unroll
            for (int v = 0; v < TN; ++v) bf[v] = *reinterpret_cast<const float4*>(sb + v * 32 * 32 + 4 * (q ^ sw));
            float a0[TM], a1[TM], b0[TN], b1[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) { a0[t] = kh ? af[t].y : af[t].x; a1[t] = kh ? af[t].w : af[t].z; }
#pragma unroll
            for (int v = 0; v < TN; ++v) { b0[v] = kh ? bf[v].y : bf[v].x; b1[v] = kh ? bf[v].w : bf[v].z; }
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int v = 0; v < TN; ++v) acc[t][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[v], acc[t][v], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int v = 0; v < TN; ++v) acc[t][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[v], acc[t][v], 0, 0, 0);
        }
        stage ^= 1;
    }
    __syncthreads();
    store_tiles<TM, TN>(g, acc, smem + wave * 32 * (32 * TN + 4), m0, n0, wm, wn, lane);
#endif
}

// ---------------------------------------------------------------------------------------------- MFMA ceiling (no memory)
// 4 independent accumulators per wave, operands from memory (random or zero): the matrix pipe's rate at the clock the chip
// sustains for THIS data (DVFS: MI355X_MICROARCH.md, zero-filled operands clock higher).
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD / 1) void mfma_peak_kernel(const float* in, float* out, int iters)
{
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { a[t] = in[(threadIdx.x * 8 + t) & 4095]; b[t] = in[(threadIdx.x * 8 + 4 + t) & 4095]; }
    for (int it = 0; it < iters; it += 4) {
        // operands change from instruction to instruction like in a GEMM (register rotation, no extra VALU work)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(t + 1) & 3], b[(t + 2) & 3], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[(t + 3) & 3], a[(t + 2) & 3], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t], a[(t + 3) & 3], acc[t], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------- host
typedef int (*gemm_fn)(const RoitrGemm*, hipStream_t);

template <typename K>
static float time_kernel(K launch, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

template <int TM, int TN, int DMA, int AB = 0>
static void run_variant(const P& p, float* ref_host, std::vector<float>& out_host, const char* name, int reps)
{
    const int nx = (p.N + 64 * TN - 1) / (64 * TN), ny = (p.M + 64 * TM - 1) / (64 * TM);
    const int T = nx * ny;
    const size_t stage_r = (size_t)(TM + TN) * 2 * 64 * LDR * 4, stage_d = (size_t)(64 * TM + 64 * TN) * 32 * 4 * 2;
    const size_t scr = (size_t)4 * 32 * (32 * TN + 4) * 4;
    size_t lds = DMA == 1 ? stage_d : DMA == 2 ? (size_t)2 * (64 * TM + 64 * TN) * SP * 4 : stage_r;
    if (lds < scr) lds = scr;
    auto kern = DMA == 1 ? (void (*)(P, int, int))gemm_d_kernel<TM, TN> : DMA == 2 ? (void (*)(P, int, int))gemm_s_kernel<TM, TN> : (void (*)(P, int, int))gemm_r_kernel<TM, TN, AB>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(p.C, 0xff, (size_t)p.M * p.ldc * 4));
    auto launch = [&]() { hipLaunchKernelGGL(kern, dim3(xcd_grid(T)), dim3(256), lds, 0, p, nx, T); };
    const float ms = time_kernel(launch, reps);
    CK(hipMemcpy(out_host.data(), p.C, (size_t)p.M * p.ldc * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    if (ref_host) for (size_t i = 0; i < (size_t)p.M * p.N; ++i) bad += memcmp(&out_host[i], &ref_host[i], 4) != 0;
    printf("  %-10s %8.4f ms  %6.1f TFLOP/s  lds %6zu  tiles %7d  mismatches %zu\n", name, ms, 2.0 * p.M * p.N * p.K / ms / 1e9, lds, T, bad);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const char* libpath = argc > 1 ? argv[1] : "roitr_amd/lib/libroitr_hip.so";
    void* h = dlopen(libpath, RTLD_NOW);
    gemm_fn lib_gemm = h ? (gemm_fn)dlsym(h, "roitr_gemm") : nullptr;
    if (!lib_gemm) fprintf(stderr, "no library gemm (%s): timing only\n", dlerror());
    const bool zero = argc > 2 && !strcmp(argv[2], "zero");
    {   // MFMA ceiling with random / zero operands
        float *din, *dout;
        std::vector<float> hin(4096);
        CK(hipMalloc(&din, 4096 * 4)); CK(hipMalloc(&dout, (size_t)256 * 8 * 256 * 4));
        for (int z = 0; z < 2; ++z) {
            uint32_t x = 777u;
            for (auto& v : hin) { x = x * 1664525u + 1013904223u; v = z ? 0.f : ((x >> 8) & 0xffff) / 32768.0f - 1.0f; }
            CK(hipMemcpy(din, hin.data(), 4096 * 4, hipMemcpyHostToDevice));
            const int iters = 20000, blocks = 256 * 2;
            auto launch = [&]() { hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(blocks), dim3(256), 0, 0, din, dout, iters); };
            const float ms = time_kernel(launch, 5);
            printf("mfma ceiling (%s operands, 2 waves/SIMD x 4 accumulators): %.1f TFLOP/s\n", z ? "zero" : "random",
                   (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 2 / ms / 1e9);
        }
        CK(hipFree(din)); CK(hipFree(dout));
    }
    struct Shape { int M, N, K; };
    const Shape shapes[] = {{319488, 768, 256}, {319488, 256, 512}, {319488, 256, 256}, {79872, 768, 256}, {79872, 256, 512}, {79872, 256, 256},
                            {39936, 256, 256}, {39936, 512, 256}, {1280000, 256, 128}, {4096, 4096, 4096}};
    const int only_shape = argc > 3 ? atoi(argv[3]) : -1;
    const char* only_var = argc > 4 ? argv[4] : "";
    int shape_i = -1;
    for (const Shape& s : shapes) {
        if (++shape_i != only_shape && only_shape >= 0) continue;
        const size_t na = (size_t)s.M * s.K, nw = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
        std::vector<float> ha(na), hw(nw), hb(s.N), hc(nc), href(nc);
        uint32_t x = 12345u;
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto& v : ha) v = zero ? 0.f : rnd();
        for (auto& v : hw) v = zero ? 0.f : rnd();
        for (auto& v : hb) v = rnd();
        float *dA, *dW, *dB, *dC;
        CK(hipMalloc(&dA, na * 4)); CK(hipMalloc(&dW, nw * 4)); CK(hipMalloc(&dB, s.N * 4)); CK(hipMalloc(&dC, nc * 4));
        CK(hipMemcpy(dA, ha.data(), na * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hw.data(), nw * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hb.data(), s.N * 4, hipMemcpyHostToDevice));
        printf("M %d N %d K %d\n", s.M, s.N, s.K);
        const int reps = 20;
        bool have_ref = false;
        if (lib_gemm) {
            RoitrGemm g; memset(&g, 0, sizeof g);
            g.M = s.M; g.N = s.N; g.K = s.K; g.A = dA; g.lda = s.K; g.W = dW; g.ldw = s.K; g.bias = dB; g.alpha = 1.0f; g.C = dC; g.ldc = s.N; g.batch = 1;
            auto launch = [&]() { if (lib_gemm(&g, 0) != 0) { fprintf(stderr, "lib gemm failed\n"); exit(1); } };
            const float ms = time_kernel(launch, reps);
            CK(hipMemcpy(href.data(), dC, nc * 4, hipMemcpyDeviceToHost));
            have_ref = true;
            printf("  %-10s %8.4f ms  %6.1f TFLOP/s\n", "library", ms, 2.0 * s.M * s.N * s.K / ms / 1e9);
        }
        P p{dA, dW, dB, dC, s.M, s.N, s.K, s.K, s.K, s.N, 0};
        float* ref = have_ref ? href.data() : nullptr;
        auto want = [&](const char* n) { return strstr(n, only_var) != nullptr; };
        if (want("R<1,1>")) run_variant<1, 1, 0>(p, ref, hc, "R<1,1>", reps);
        if (want("R<1,2>")) run_variant<1, 2, 0>(p, ref, hc, "R<1,2>", reps);
        if (want("R<2,1>")) run_variant<2, 1, 0>(p, ref, hc, "R<2,1>", reps);
        if (want("R<2,2>")) run_variant<2, 2, 0>(p, ref, hc, "R<2,2>", reps);
        if (want("R<1,1>ab1")) run_variant<1, 1, 0, 1>(p, nullptr, hc, "R<1,1>ab1", reps);
        if (want("R<1,1>ab2")) run_variant<1, 1, 0, 2>(p, nullptr, hc, "R<1,1>ab2", reps);
        if (want("R<1,1>ab3")) run_variant<1, 1, 0, 3>(p, nullptr, hc, "R<1,1>ab3", reps);
        if (want("R<1,1>ab4")) run_variant<1, 1, 0, 4>(p, nullptr, hc, "R<1,1>ab4", reps);
        if (want("R<2,2>ab1")) run_variant<2, 2, 0, 1>(p, nullptr, hc, "R<2,2>ab1", reps);
        if (want("R<2,2>ab2")) run_variant<2, 2, 0, 2>(p, nullptr, hc, "R<2,2>ab2", reps);
        if (want("R<2,2>ab3")) run_variant<2, 2, 0, 3>(p, nullptr, hc, "R<2,2>ab3", reps);
        if (want("R<2,2>ab4")) run_variant<2, 2, 0, 4>(p, nullptr, hc, "R<2,2>ab4", reps);
        if (want("R<1,1>ab5")) run_variant<1, 1, 0, 5>(p, nullptr, hc, "R<1,1>ab5", reps);
        if (want("R<1,1>ab6")) run_variant<1, 1, 0, 6>(p, nullptr, hc, "R<1,1>ab6", reps);
        if (want("R<1,1>ab7")) run_variant<1, 1, 0, 7>(p, nullptr, hc, "R<1,1>ab7", reps);
        if (want("R<1,1>ab9")) run_variant<1, 1, 0, 9>(p, ref, hc, "R<1,1>ab9", reps);
        if (want("R<1,2>ab9")) run_variant<1, 2, 0, 9>(p, ref, hc, "R<1,2>ab9", reps);
        if (want("R<2,1>ab9")) run_variant<2, 1, 0, 9>(p, ref, hc, "R<2,1>ab9", reps);
        if (want("R<2,2>ab9")) run_variant<2, 2, 0, 9>(p, ref, hc, "R<2,2>ab9", reps);
        if (want("R<1,1>ab8")) run_variant<1, 1, 0, 8>(p, nullptr, hc, "R<1,1>ab8", reps);
        if (want("R<2,2>ab8")) run_variant<2, 2, 0, 8>(p, nullptr, hc, "R<2,2>ab8", reps);
        if (want("R<2,2>ab5")) run_variant<2, 2, 0, 5>(p, nullptr, hc, "R<2,2>ab5", reps);
        if (want("R<2,2>ab6")) run_variant<2, 2, 0, 6>(p, nullptr, hc, "R<2,2>ab6", reps);
        if (want("R<2,2>ab7")) run_variant<2, 2, 0, 7>(p, nullptr, hc, "R<2,2>ab7", reps);
        if (want("S<1,1>")) run_variant<1, 1, 2>(p, ref, hc, "S<1,1>", reps);
        if (want("S<1,2>")) run_variant<1, 2, 2>(p, ref, hc, "S<1,2>", reps);
        if (want("S<2,1>")) run_variant<2, 1, 2>(p, ref, hc, "S<2,1>", reps);
        if (want("S<2,2>")) run_variant<2, 2, 2>(p, ref, hc, "S<2,2>", reps);
        if (want("D<1,1>")) run_variant<1, 1, 1>(p, ref, hc, "D<1,1>", reps);
        if (want("D<1,2>")) run_variant<1, 2, 1>(p, ref, hc, "D<1,2>", reps);
        if (want("D<2,1>")) run_variant<2, 1, 1>(p, ref, hc, "D<2,1>", reps);
        if (want("D<2,2>")) run_variant<2, 2, 1>(p, ref, hc, "D<2,2>", reps);
        CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    return 0;
}
