// Correspondence-quality evaluators on device, batched over pairs.
//
// Reference: lib/loss.py:169-213 (Evaluator: PIR = evaluate_coarse, IR = evaluate_fine, eval_acceptance_overlap /
// eval_acceptance_radius) and registration/benchmark_utils.py:69-77 (get_inlier_ratio_correspondence): the numbers the
// "Inlier Ratio within 0.1 pp" line of the north star is stated in.  Counts are returned; the ratio (count / number of
// correspondences, 0 for an empty set: lib/loss.py:199-200) is formed by the caller.
#include "common.h"
#include "roitr_engine.h"

namespace {

// one block per pair: inliers = #{ c : || R s_c + t - g_c || < radius }   (lib/loss.py:202-204)
__global__ __launch_bounds__(256) void inlier_count_kernel(const int* __restrict__ starts, const float* __restrict__ src, const float* __restrict__ tgt,
                                                           const float* __restrict__ rot, const float* __restrict__ trans, float radius,
                                                           int* __restrict__ inliers)
{
    __shared__ int wsum[4];
    const int b = blockIdx.x;
    const int s = starts[b], e = starts[b + 1];
    const float* R = rot + (size_t)b * 9;
    const float* t = trans + (size_t)b * 3;
    const float r00 = R[0], r01 = R[1], r02 = R[2], r10 = R[3], r11 = R[4], r12 = R[5], r20 = R[6], r21 = R[7], r22 = R[8];
    const float t0 = t[0], t1 = t[1], t2 = t[2];
    int local = 0;
    for (int c = s + threadIdx.x; c < e; c += 256) {
        const float x = src[(size_t)c * 3], y = src[(size_t)c * 3 + 1], z = src[(size_t)c * 3 + 2];
        // src @ rot.T + trans.T: row i of rot dotted with the point, fp32 FMA chain in k order
        const float px = fmaf(z, r02, fmaf(y, r01, x * r00)) + t0;
        const float py = fmaf(z, r12, fmaf(y, r11, x * r10)) + t1;
        const float pz = fmaf(z, r22, fmaf(y, r21, x * r20)) + t2;
        const float dx = tgt[(size_t)c * 3] - px, dy = tgt[(size_t)c * 3 + 1] - py, dz = tgt[(size_t)c * 3 + 2] - pz;
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        local += d < radius ? 1 : 0;
    }
    local = (int)wave_sum((float)local);   // exact: < 2^24 per wave
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) inliers[b] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one block per pair: hits = #{ i < n_corr : (tgt_corr[i], src_corr[i]) is a GT node pair with overlap > thr }
// (lib/loss.py:176-191: gt map filled from the filtered list, then indexed by the predicted pairs; duplicates count)
__global__ __launch_bounds__(256) void coarse_hits_kernel(int num_corr, const int* __restrict__ n_corr, const int* __restrict__ tgt_corr,
                                                          const int* __restrict__ src_corr, int gt_cap, const int* __restrict__ gt_idx,
                                                          const float* __restrict__ gt_overlaps, const int* __restrict__ gt_count, float thr,
                                                          int* __restrict__ hits)
{
    __shared__ int wsum[4];
    const int b = blockIdx.x;
    const int nc = min(n_corr[b], num_corr), ng = min(gt_count[b], gt_cap);
    const int* gi = gt_idx + (size_t)b * gt_cap * 2;
    const float* go = gt_overlaps + (size_t)b * gt_cap;
    int local = 0;
    for (int i = threadIdx.x; i < nc; i += 256) {
        const int t = tgt_corr[(size_t)b * num_corr + i], s = src_corr[(size_t)b * num_corr + i];
        int hit = 0;
        for (int j = 0; j < ng; ++j) hit |= (gi[2 * j] == t && gi[2 * j + 1] == s && go[j] > thr) ? 1 : 0;
        local += hit;
    }
    local = (int)wave_sum((float)local);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) hits[b] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

}  // namespace

extern "C" int roitr_inlier_counts(int pairs, const int* starts, const float* src_pts, const float* tgt_pts, const float* rot,
                                   const float* trans, float radius, int* inliers, hipStream_t stream)
{
    if (pairs <= 0) return ROITR_OK;
    if (!starts || !rot || !trans || !inliers) return ROITR_ERR_ARG;
    inlier_count_kernel<<<pairs, 256, 0, stream>>>(starts, src_pts, tgt_pts, rot, trans, radius, inliers);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_coarse_hits(int pairs, int num_corr, const int* n_corr, const int* tgt_corr, const int* src_corr, int gt_cap,
                                 const int* gt_idx, const float* gt_overlaps, const int* gt_count, float acceptance_overlap, int* hits,
                                 hipStream_t stream)
{
    if (pairs <= 0) return ROITR_OK;
    if (!n_corr || !tgt_corr || !src_corr || !gt_idx || !gt_overlaps || !gt_count || !hits) return ROITR_ERR_ARG;
    coarse_hits_kernel<<<pairs, 256, 0, stream>>>(num_corr, n_corr, tgt_corr, src_corr, gt_cap, gt_idx, gt_overlaps, gt_count,
                                                  acceptance_overlap, hits);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}
