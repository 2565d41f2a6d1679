// Probe for the short-K fp32 layers (M = 5.12 M rows, N = K = 64: 1.07 ms = 2.45 TB/s, 39 TFLOP/s inside the engine).
// Which part of the 64 x 64 tile kernel is that launch waiting for?  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I roitr_amd/csrc scripts/micro/gemm_k64_probe.hip -o /tmp/gemm_probe && /tmp/gemm_probe
// Modes: 0 = the engine kernel's staging (two 32-wide slabs one after the other, 4 lanes x 2 x 16 B per row and slab),
//        1 = mode 0 without the C stores, 2 = mode 0 with the A rows read from a 16 KB L2-resident buffer (no HBM reads),
//        3 = whole A and W tiles fetched at once, every wave instruction reading 1 KB of contiguous rows,
//        4 = mode 3 with two row tiles per block (BM = 128, the W image staged once), 5 = plain float4 copy A -> C (HBM reference),
//        6 = mode 0 without the XCD-aware tile order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 64, BN = 64, BK = 32, LDR = 20, K = 64, N = 64;

__device__ __forceinline__ int xcd_block_id(int n) { const int b = blockIdx.x; return (b & 7) * ((n + 7) >> 3) + (b >> 3); }

template <int MODE>
__global__ __launch_bounds__(256) void probe_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C, int M, int T,
                                                    const float* __restrict__ small)
{
    constexpr int RT = MODE == 4 ? 2 : 1;   // row tiles per block
    constexpr bool WHOLE = MODE == 3 || MODE == 4;                 // both K-slabs of the tile resident at once
    constexpr int ASZ = WHOLE ? RT * 2 * 2 * BM * LDR : 2 * BM * LDR;   // floats
    constexpr int BSZ = WHOLE ? 2 * 2 * BN * LDR : 2 * BN * LDR;
    __shared__ __attribute__((aligned(16))) float smem[ASZ + BSZ];  // 20 KB like the engine kernel, 40 / 60 KB for modes 3 / 4
    float* As = smem;
    float* Bs = smem + ASZ;
    float* Cs = smem;   // the C tile is parked over the operand images once every wave is past its MFMAs
    const int tile = MODE == 6 ? blockIdx.x : xcd_block_id(T);
    if (tile >= T) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = tile * BM * RT;
    const int kh = lane >> 5, ml = lane & 31;
    f32x16 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    if (MODE <= 2 || MODE == 6) {
        const int r = tid >> 2, kq = (tid & 3) * 8, kf = (tid & 3) * 4;
        const float* arow = MODE == 2 ? small + (size_t)r * K : A + (size_t)min(m0 + r, M - 1) * K;
        const float* wrow = W + (size_t)r * K;
        float av[8], wv[8];
        auto ld8 = [&](const float* p, float (&d)[8]) {
            const float4 x = *reinterpret_cast<const float4*>(p), y = *reinterpret_cast<const float4*>(p + 16);
            d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w; d[4] = y.x; d[5] = y.y; d[6] = y.z; d[7] = y.w;
        };
        ld8(arow + kf, av); ld8(wrow + kf, wv);
        const float4* ar = reinterpret_cast<const float4*>(As + (kh * BM + wm * 32 + ml) * LDR);
        const float4* br = reinterpret_cast<const float4*>(Bs + (kh * BN + wn * 32 + ml) * LDR);
        float4* aw0 = reinterpret_cast<float4*>(As + (0 * BM + r) * LDR + (kq >> 1));
        float4* aw1 = reinterpret_cast<float4*>(As + (1 * BM + r) * LDR + (kq >> 1));
        float4* bw0 = reinterpret_cast<float4*>(Bs + (0 * BN + r) * LDR + (kq >> 1));
        float4* bw1 = reinterpret_cast<float4*>(Bs + (1 * BN + r) * LDR + (kq >> 1));
        for (int k0 = 0; k0 < K; k0 += BK) {
            __syncthreads();
            *aw0 = make_float4(av[0], av[2], av[4], av[6]); *aw1 = make_float4(av[1], av[3], av[5], av[7]);
            *bw0 = make_float4(wv[0], wv[2], wv[4], wv[6]); *bw1 = make_float4(wv[1], wv[3], wv[5], wv[7]);
            __syncthreads();
            if (k0 + BK < K) { ld8(arow + k0 + BK + kf, av); ld8(wrow + k0 + BK + kf, wv); }
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
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    } else {
        // whole tiles at once: load i of a thread reads row 16 i + tid / 16, floats 4 (tid % 16) .. +3 -> a wave instruction
        // covers four consecutive 256-byte rows = 1 KB contiguous
        const int rr = tid >> 4, k4 = (tid & 15) * 4;
        float4 a[RT][4], w[4];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[t][i] = *reinterpret_cast<const float4*>(A + (size_t)min(m0 + t * BM + i * 16 + rr, M - 1) * K + k4);
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = *reinterpret_cast<const float4*>(W + (size_t)(i * 16 + rr) * K + k4);
        const int s = k4 >> 5, kk0 = (k4 & 31) >> 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + rr;
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                float* pa = As + t * (2 * 2 * BM * LDR) + s * (2 * BM * LDR) + row * LDR + kk0;
                *reinterpret_cast<float2*>(pa) = make_float2(a[t][i].x, a[t][i].z);
                *reinterpret_cast<float2*>(pa + BM * LDR) = make_float2(a[t][i].y, a[t][i].w);
            }
            float* pb = Bs + s * (2 * BN * LDR) + row * LDR + kk0;
            *reinterpret_cast<float2*>(pb) = make_float2(w[i].x, w[i].z);
            *reinterpret_cast<float2*>(pb + BN * LDR) = make_float2(w[i].y, w[i].w);
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const float4* ar = reinterpret_cast<const float4*>(As + t * (2 * 2 * BM * LDR) + (kh * BM + wm * 32 + ml) * LDR);
            const float4* br = reinterpret_cast<const float4*>(Bs + (kh * BN + wn * 32 + ml) * LDR);
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                float4 af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { af[i] = ar[sl * (2 * BM * LDR) / 4 + i]; bf[i] = br[sl * (2 * BN * LDR) / 4 + i]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[i].x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[i].y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[i].z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[i].w, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // wide store through LDS (the engine kernel's epilogue)
    constexpr int TP = BN + 4;
    if (WHOLE) __syncthreads();
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        if (t) __syncthreads();
        const int col = wn * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rl = wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            Cs[rl * TP + col] = acc[t][i];
        }
        __syncthreads();
        const int c4 = (tid & 15) * 4;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int rl = pass * 16 + (tid >> 4);
            const int row = m0 + t * BM + rl;
            const float4 v = *reinterpret_cast<const float4*>(Cs + rl * TP + c4);
            if (MODE == 1) { if (v.x == 123456.789f && row < M) C[(size_t)row * N + c4] = v.x; }
            else if (row < M) *reinterpret_cast<float4*>(C + (size_t)row * N + c4) = v;
        }
    }
}

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ a, float4* __restrict__ c, long n4)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) c[i] = a[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
void run(const float* A, const float* W, float* C, int M, const float* small, const std::vector<float>& hA, const std::vector<float>& hW)
{
    const int RT = MODE == 4 ? 2 : 1;
    const int T = (M + BM * RT - 1) / (BM * RT);
    const int grid = MODE == 6 ? T : ((T + 7) / 8) * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(C, 0, (size_t)M * N * 4));
    for (int w = 0; w < 2; ++w) probe_kernel<MODE><<<grid, 256>>>(A, W, C, M, T, small);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int it = 10;
    for (int i = 0; i < it; ++i) probe_kernel<MODE><<<grid, 256>>>(A, W, C, M, T, small);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
    double err = -1;
    if (MODE == 0 || MODE == 3 || MODE == 4 || MODE == 6) {   // sanity: rows 0..63 and the last 64 against float64
        std::vector<float> hc((size_t)64 * N);
        err = 0;
        for (int blk = 0; blk < 2; ++blk) {
            const int r0 = blk ? M - 64 : 0;
            CK(hipMemcpy(hc.data(), C + (size_t)r0 * N, hc.size() * 4, hipMemcpyDeviceToHost));
            for (int r = 0; r < 64; ++r)
                for (int n = 0; n < N; ++n) {
                    double s = 0; for (int k = 0; k < K; ++k) s += (double)hA[(size_t)(r0 + r) * K + k] * hW[(size_t)n * K + k];
                    err = fmax(err, fabs(s - hc[(size_t)r * N + n]));
                }
        }
    }
    const double bytes = (double)M * (K + N) * 4;
    printf("mode %d: %.3f ms  %.2f TB/s  %.1f TFLOP/s  max err %.2e\n", MODE, ms, bytes / ms / 1e9, 2.0 * M * N * K / ms / 1e9, err);
}

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 5120000;
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hW) v = rnd();
    float *A, *W, *C, *small;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&small, 64 * K * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(small, hA.data(), 64 * K * 4, hipMemcpyHostToDevice));
    run<0>(A, W, C, M, small, hA, hW);
    run<1>(A, W, C, M, small, hA, hW);
    run<2>(A, W, C, M, small, hA, hW);
    run<3>(A, W, C, M, small, hA, hW);
    run<4>(A, W, C, M, small, hA, hW);
    run<6>(A, W, C, M, small, hA, hW);
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const long n4 = (long)M * K / 4;
        for (int g : {2048, 8192, 65536}) {
            copy_kernel<<<g, 256>>>((const float4*)A, (float4*)C, n4);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) copy_kernel<<<g, 256>>>((const float4*)A, (float4*)C, n4);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
            printf("copy grid %d: %.3f ms  %.2f TB/s\n", g, ms, (double)M * K * 8 / ms / 1e9);
        }
    }
    return 0;
}
