// Point-pair features for given neighbour indices: lib/utils.py:358-389 calc_ppf_gpu (reference).
// (The hot path gets its PPF fused into the kNN pass, pointops_knn.hip; this standalone kernel serves
// callers that already hold group indices, and the operator-level API roitr_amd.ops.calc_ppf.)
// HBM-bound: one lane per (centre, neighbour) pair, float4 store; centre data is a wave-wide
// broadcast when k >= 64 and L1-resident otherwise.
#include "common.h"

namespace {
__global__ void ppf_kernel(long total, int k, const float* __restrict__ c_xyz, const float* __restrict__ c_n,
                           const float* __restrict__ r_xyz, const float* __restrict__ r_n, const int* __restrict__ grp,
                           float4* __restrict__ out)
{
    for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += gridDim.x * 256L) {
        const long m = t / k;
        const long g = grp[t];
        const float cx = c_xyz[m * 3], cy = c_xyz[m * 3 + 1], cz = c_xyz[m * 3 + 2];
        const float nx = c_n[m * 3], ny = c_n[m * 3 + 1], nz = c_n[m * 3 + 2];
        const float px = r_n[g * 3], py = r_n[g * 3 + 1], pz = r_n[g * 3 + 2];
        const float4 o = roitr_ppf4(cx, cy, cz, nx, ny, nz, r_xyz[g * 3], r_xyz[g * 3 + 1], r_xyz[g * 3 + 2], px, py, pz);
        out[t] = o;
    }
}
}  // namespace

// centres (m,3)+(m,3) normals; reference cloud (n,3)+(n,3); grp (m,k) int32 into the reference cloud;
// out (m,k,4).
extern "C" int roitr_calc_ppf(int m, int k, const float* centre_xyz, const float* centre_normals, const float* ref_xyz,
                              const float* ref_normals, const int* group_idx, float* out, hipStream_t stream)
{
    const long total = (long)m * k;
    if (total <= 0) return ROITR_OK;
    int blocks = div_up(total, 256);
    if (blocks > 8192) blocks = 8192;
    ppf_kernel<<<blocks, 256, 0, stream>>>(total, k, centre_xyz, centre_normals, ref_xyz, ref_normals, group_idx, (float4*)out);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}
