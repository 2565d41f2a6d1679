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
// error.cpp: raise a kernel's dynamic-LDS limit on the current device (once per kernel and device, checked)
int roitr_grant_dynamic_lds(const void* kernel, int bytes);
#define ROITR_GRANT_LDS(kernel, bytes)                                                  \
    do {                                                                                \
        const int g__ = roitr_grant_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)(bytes)); \
        if (g__ != ROITR_OK) return g__;                                                \
    } while (0)

static inline int div_up(long a, long b) { return (int)((a + b - 1) / b); }

// Squared distance in the arithmetic form shared with oracle/pointops_ref.c:
// fmaf(dz,dz, fmaf(dy,dy, dx*dx)) -- the nvcc --fmad=true contraction of the reference's
// `dx*dx + dy*dy + dz*dz` (knnquery_cuda_kernel.cu:96, sampling_cuda_kernel.cu:54).
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}

// Point-pair features (lib/utils.py:358-389 calc_ppf_gpu): [ |d|, angle(n1, d), angle(n2, d), angle(n1, n2) ], angles / pi with
// angle(a, b) = atan2(|a x b|, a . b) in [0, pi].  One definition for every kernel that emits PPFs (the fused kNN kernels, ppf.hip):
// a pair's features must not depend on which kernel served its query.  Round 5: the three atan2f (OCML: ~60 instructions each plus an
// IEEE division by pi and correctly rounded square roots -- 248 VALU instructions per feature, the largest single item of the lane
// kernels) are a first-quadrant reduction t = min / max (v_rcp_f32), a degree-8 polynomial in t^2 (Chebyshev fit of atan(t) / t on
// [0, 1]: 1.1e-7 rad in fp32 Horner form) and two reflections; v_sqrt_f32 for the norms; 1 / pi as a factor.  Against float64:
// <= 2.5e-7 on the angles / pi (tests/test_pointops_gpu.py, tolerance of the PPF tests: 2e-6).  atan2(0, +0) = 0 like torch (the dot
// product starts from +0, so x is never -0); a zero vector gives 0.
__device__ __forceinline__ float roitr_angle_over_pi(float y, float x)   // y = |a x b| >= 0, x = a . b
{
    const float ax = fabsf(x);
    const float lo = fminf(ax, y), hi = fmaxf(ax, y);
    float t = lo * __builtin_amdgcn_rcpf(hi);
    t = lo > 0.f ? fminf(t, 1.0f) : 0.f;          // hi = 0 (both zero) or a flushed denormal: 0 * inf
    const float s = t * t;
    float p = 0.0028340641874819994f;
    p = __fmaf_rn(p, s, -0.016005029901862144f);
    p = __fmaf_rn(p, s, 0.042587608098983765f);
    p = __fmaf_rn(p, s, -0.07495445758104324f);
    p = __fmaf_rn(p, s, 0.10636754333972931f);
    p = __fmaf_rn(p, s, -0.14202570915222168f);
    p = __fmaf_rn(p, s, 0.19992484152317047f);
    p = __fmaf_rn(p, s, -0.3333306610584259f);
    p = __fmaf_rn(p, s, 1.0f);
    p *= t;                                        // atan(t), t in [0, 1]
    p = y > ax ? 1.57079632679489662f - p : p;     // first octant -> first quadrant
    p = x < 0.f ? 3.14159265358979323846f - p : p; // second quadrant
    return p * 0.318309886183790672f;
}
__device__ __forceinline__ float roitr_angle3(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dt = 0.0f + ax * bx + ay * by + az * bz;  // torch.sum starts from +0: keeps atan2(0, +0) = 0
    const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
    return roitr_angle_over_pi(__builtin_amdgcn_sqrtf(cx * cx + cy * cy + cz * cz), dt);
}
// centre c (normal cn) vs neighbour p (normal pn)
__device__ __forceinline__ float4 roitr_ppf4(float cx, float cy, float cz, float cnx, float cny, float cnz, float px, float py, float pz,
                                             float pnx, float pny, float pnz)
{
    const float dx = px - cx, dy = py - cy, dz = pz - cz;
    float4 o;
    o.x = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
    o.y = roitr_angle3(cnx, cny, cnz, dx, dy, dz);
    o.z = roitr_angle3(pnx, pny, pnz, dx, dy, dz);
    o.w = roitr_angle3(cnx, cny, cnz, pnx, pny, pnz);
    return o;
}

// Segment (cloud) of element i under cumulative offsets off[0..n): the first c with i < off[c] + c * stride + bias, clamped to
// n - 1.  A binary search: the batched engine has up to ~1000 clouds per call, a linear walk per thread shows up in profiles.
__device__ __forceinline__ int segment_of(int i, const int* __restrict__ off, int n, int stride = 0, int bias = 0)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (i < off[mid] + mid * stride + bias) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Sum over the 16 lanes of a DPP row (result in every lane of the row).
template <int CTRL>
__device__ __forceinline__ float row_dpp_add(float v)
{
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, result in every lane: quad xor 1, quad xor 2, half mirror, row mirror
__device__ __forceinline__ float row_allsum(float v)
{
    v = row_dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = row_dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = row_dpp_add<0x141>(v);  // row_half_mirror
    v = row_dpp_add<0x140>(v);  // row_mirror
    return v;
}

// Sum 16 values per lane over the 16 lanes of a DPP row "transposed": lane i ends with the row total of ONE value,
//     value index  row16_slot(i) = (bit2(i) << 3) | (bit0(i) << 2) | (bit1(i) << 1) | bit3(i),
// in 15 exchange-and-add steps (8 + 4 + 2 + 1) instead of 16 x 4: a lane keeps the half of its values that matches one of its
// lane bits and receives the partner's partial sums for that half.  The first exchange uses row_half_mirror (partner = lane ^ 7:
// nothing is resolved yet, any partner with the other bit 2 will do), then quad_perm lane ^ 1, lane ^ 2 and row_ror:8 (lane ^ 8).
template <int CTRL> __device__ __forceinline__ float dpp_get(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ int row16_slot(int i) { return ((i >> 2) & 1) << 3 | (i & 1) << 2 | ((i >> 1) & 1) << 1 | ((i >> 3) & 1); }
__device__ __forceinline__ float row16_transpose_sum(const float (&v)[16], int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    float w[8], u[4], t[2];
#pragma unroll
    for (int m = 0; m < 8; ++m) w[m] = (b2 ? v[8 + m] : v[m]) + dpp_get<0x141>(b2 ? v[m] : v[8 + m]);   // row_half_mirror
#pragma unroll
    for (int m = 0; m < 4; ++m) u[m] = (b0 ? w[4 + m] : w[m]) + dpp_get<0xB1>(b0 ? w[m] : w[4 + m]);    // quad_perm [1,0,3,2]
#pragma unroll
    for (int m = 0; m < 2; ++m) t[m] = (b1 ? u[2 + m] : u[m]) + dpp_get<0x4E>(b1 ? u[m] : u[2 + m]);    // quad_perm [2,3,0,1]
    return (b3 ? t[1] : t[0]) + dpp_get<0x128>(b3 ? t[0] : t[1]);                                        // row_ror:8
}

// gfx950 lane-swap adds: fold the two halves / the row pairs of a wave while every lane keeps one of two values
typedef unsigned la_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float swap32_sum(float a, float b)   // lanes < 32: sum of a over (l, l + 32); lanes >= 32: the same for b
{
    const la_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float swap16_sum(float a, float b)   // even rows: sum of a over (row, row + 1); odd rows: the same for b
{
    const la_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// XCD-aware block id.  The dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md, workgroup dispatch), each XCD
// with a private 4 MiB L2: consecutive blocks that share gathered rows would each fetch them into a different L2.
// This remap hands every XCD one contiguous eighth of the logical block range.  Launch xcd_grid(n) blocks and skip
// logical ids >= n (padding of the rounded-up grid).
__host__ __device__ __forceinline__ int xcd_grid(int n) { return ((n + 7) >> 3) << 3; }
__device__ __forceinline__ int xcd_block_id(int n)
{
    const int b = blockIdx.x;
    return (b & 7) * ((n + 7) >> 3) + (b >> 3);
}

// The same map for a launch whose LIVE block count n is only known on the device (n <= the count the grid was sized for): blocks
// past the live range return -1 (the plain form would fold them onto the next XCD's tiles).
__device__ __forceinline__ int xcd_block_id_live(int n)
{
    const int b = blockIdx.x, per = (n + 7) >> 3;
    if ((b >> 3) >= per) return -1;
    const int t = (b & 7) * per + (b >> 3);
    return t < n ? t : -1;
}

// Wave-wide reductions on the DPP cross-lane path (no LDS traffic, unlike ds_bpermute-backed __shfl_xor for the 16/32
// strides): row_shr 1/2/4/8 build the 16-lane row result in lane 15 of each row, row_bcast:15 / row_bcast:31 fold the
// four rows, lane 63 holds the wave result and v_readlane broadcasts it.  Lanes without a source read `old`
// (bound_ctrl = 0): the identity for the sum, the lane's own value for the idempotent max.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_f32(float v)
{
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_f32(float v)
{
    const int i = __float_as_int(v);
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xf, false)));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v = dpp_add_f32<0x111, 0xf>(v); v = dpp_add_f32<0x112, 0xf>(v); v = dpp_add_f32<0x114, 0xf>(v); v = dpp_add_f32<0x118, 0xf>(v);
    v = dpp_add_f32<0x142, 0xa>(v); v = dpp_add_f32<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v)
{
    v = dpp_max_f32<0x111, 0xf>(v); v = dpp_max_f32<0x112, 0xf>(v); v = dpp_max_f32<0x114, 0xf>(v); v = dpp_max_f32<0x118, 0xf>(v);
    v = dpp_max_f32<0x142, 0xa>(v); v = dpp_max_f32<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
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
