// Attainable fp32 MFMA rate on this GPU (no memory traffic): every wave runs NACC independent
// v_mfma_f32_32x32x2_f32 accumulator chains.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b)
{
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu)
{
    const int blocks = 256 * blocks_per_cu, iters = 4000;
    float* d; hipMalloc(&d, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(d, 100, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
    printf("NACC %d blocks/CU %d: %.2f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(d);
}
int main()
{
    run<1>(1); run<1>(2); run<1>(4); run<1>(8); run<2>(1); run<2>(2); run<4>(1); run<4>(2);
    return 0;
}
