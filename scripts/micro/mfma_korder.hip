// Probe: is v_mfma_f32_16x16x4_f32 the same k-ordered fp32 fma chain as v_mfma_f32_32x32x2_f32 (and as a scalar fmaf loop)?
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_korder.hip -o /tmp/mfma_korder && /tmp/mfma_korder
// Prints the number of output elements whose bits differ from the scalar chain, for both instructions, on operands with a wide
// exponent spread (so that a different summation order or an unfused product shows up).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int K = 64;
__global__ void probe(const float* A, const float* B, float* c32, float* c16, float* cref)
{
    const int l = threadIdx.x;
    f16v acc32 = {0};
    for (int k = 0; k < K; k += 2) acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l % 32) * K + k + l / 32], B[(k + l / 32) * 32 + l % 32], acc32, 0, 0, 0);
    for (int v = 0; v < 16; ++v) c32[((v / 4) * 8 + (l / 32) * 4 + v % 4) * 32 + l % 32] = acc32[v];
    f4v acc16 = {0};
    for (int k = 0; k < K; k += 4) acc16 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l % 16) * K + k + l / 16], B[(k + l / 16) * 32 + l % 16], acc16, 0, 0, 0);
    for (int v = 0; v < 4; ++v) c16[((l / 16) * 4 + v) * 32 + l % 16] = acc16[v];
    for (int e = l; e < 32 * 32; e += 64) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = __builtin_fmaf(A[(e / 32) * K + k], B[k * 32 + e % 32], acc);
        cref[e] = acc;
    }
}
int main()
{
    float hA[32 * K], hB[K * 32];
    srand(7);
    for (float& x : hA) x = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 24 - 12);
    for (float& x : hB) x = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 24 - 12);
    float *A, *B, *c32, *c16, *cr;
    (void)hipMalloc(&A, sizeof(hA)); (void)hipMalloc(&B, sizeof(hB)); (void)hipMalloc(&c32, 4096); (void)hipMalloc(&c16, 4096); (void)hipMalloc(&cr, 4096);
    (void)hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(B, hB, sizeof(hB), hipMemcpyHostToDevice);
    (void)hipMemset(c16, 0, 4096);
    probe<<<1, 64>>>(A, B, c32, c16, cr);
    float h32[1024], h16[1024], hr[1024];
    (void)hipMemcpy(h32, c32, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(h16, c16, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(hr, cr, 4096, hipMemcpyDeviceToHost);
    int d32 = 0, d16 = 0;
    for (int e = 0; e < 1024; ++e) d32 += memcmp(&h32[e], &hr[e], 4) != 0;
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) d16 += memcmp(&h16[r * 32 + c], &hr[r * 32 + c], 4) != 0;
    printf("mfma_korder: 32x32x2 differs from the scalar fma chain in %d of 1024 elements; 16x16x4 in %d of 256\n", d32, d16);
    return 0;
}
