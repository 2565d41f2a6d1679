// Shared helpers for the gfx950 kernels of libroitr_hip.so (wave64 only; no CUDA dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ROITR_OK 0
#define ROITR_ERR_ARG 1
#define ROITR_ERR_HIP 2
#define ROITR_ERR_UNSUPPORTED 3

#define ROITR_LAUNCH_CHECK()                                  \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) {                              \
            roitr_set_error(hipGetErrorString(e__), __FILE__, __LINE__); \
            return ROITR_ERR_HIP;                             \
        }                                                     \
    } while (0)

#define ROITR_HIP(call)                                       \
    do {                                                      \
        hipError_t e__ = (call);                              \
        if (e__ != hipSuccess) {                              \
            roitr_set_error(hipGetErrorString(e__), __FILE__, __LINE__); \
            return ROITR_ERR_HIP;                             \
        }                                                     \
    } while (0)

void roitr_set_error(const char* msg, const char* file, int line);

static inline int div_up(long a, long b) { return (int)((a + b - 1) / b); }

// Squared distance in the arithmetic form shared with oracle/pointops_ref.c:
// fmaf(dz,dz, fmaf(dy,dy, dx*dx)) -- the nvcc --fmad=true contraction of the reference's
// `dx*dx + dy*dy + dz*dz` (knnquery_cuda_kernel.cu:96, sampling_cuda_kernel.cu:54).
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}
