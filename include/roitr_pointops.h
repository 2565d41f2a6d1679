/*
 * roitr_pointops.h -- C ABI of libroitr_hip.so, part 1: the `pointops` operator boundary.
 *
 * Boundary B1 of SURVEY.md 8(b).  Plain pointers and sizes only; every pointer is a DEVICE pointer
 * to contiguous fp32 / int32 memory owned by the caller; kernels write in place.
 *
 * Two families:
 *  (1) the reference's own `extern "C"` launcher names and signatures, unchanged (void return,
 *      legacy default stream) -- a maintainer can link the reference's *_cuda.cpp ATen wrappers
 *      against this library instead of its .cu objects;
 *  (2) `roitr_*` variants of the same operators that add a hipStream_t, return an int status
 *      (0 = ok; roitr_last_error() has the text) and, for kNN, expose the grid-accelerated search
 *      and the fused queryandgroup + point-pair-feature outputs.
 *
 * All citations are into /root/reference/cpp_wrappers/pointops/src unless stated otherwise.
 */
#ifndef ROITR_POINTOPS_H
#define ROITR_POINTOPS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* roitr_stream_t; /* == hipStream_t; NULL = legacy default stream */

/* ---- diagnostics ---------------------------------------------------------------------------- */
const char* roitr_last_error(void);
int roitr_abi_version(void);

/* ---- (1) reference launcher names, verbatim ------------------------------------------------- */

/* sampling/sampling_cuda_kernel.h:13.  `n` = longest cloud (selects the tie-break block size,
 * cuda_utils.h:11-14); `tmp` (total points) must arrive filled with 1e10 (functions/pointops.py:22);
 * idx[new_offset[i-1]] = offset[i-1] (first point of each cloud is always selected). */
void furthestsampling_cuda_launcher(int b, int n, const float* xyz, const int* offset, const int* new_offset,
                                    float* tmp, int* idx);

/* knnquery/knnquery_cuda_kernel.h:13.  idx (m, nsample) int32 ascending by distance, dist2 SQUARED
 * distances; nsample <= 100 (knnquery_cuda_kernel.cu:86); slots beyond the cloud size keep
 * (offset_start, 1e10). */
void knnquery_cuda_launcher(int m, int nsample, const float* xyz, const float* new_xyz, const int* offset,
                            const int* new_offset, int* idx, float* dist2);

/* grouping/grouping_cuda_kernel.h:13-14 */
void grouping_forward_cuda_launcher(int m, int nsample, int c, const float* input, const int* idx, float* output);
void grouping_backward_cuda_launcher(int m, int nsample, int c, const float* grad_output, const int* idx, float* grad_input);
/* interpolation/interpolation_cuda_kernel.h:13-14 (output must arrive zeroed, pointops.py:199) */
void interpolation_forward_cuda_launcher(int n, int c, int k, const float* input, const int* idx, const float* weight, float* output);
void interpolation_backward_cuda_launcher(int n, int c, int k, const float* grad_output, const int* idx, const float* weight, float* grad_input);
/* subtraction/subtraction_cuda_kernel.h:13-14 */
void subtraction_forward_cuda_launcher(int n, int nsample, int c, const float* input1, const float* input2, const int* idx, float* output);
void subtraction_backward_cuda_launcher(int n, int nsample, int c, const int* idx, const float* grad_output, float* grad_input1, float* grad_input2);
/* aggregation/aggregation_cuda_kernel.h:13-14 (output must arrive zeroed, pointops.py:146) */
void aggregation_forward_cuda_launcher(int n, int nsample, int c, int w_c, const float* input, const float* position, const float* weight, const int* idx, float* output);
void aggregation_backward_cuda_launcher(int n, int nsample, int c, int w_c, const float* input, const float* position, const float* weight, const int* idx, const float* grad_output, float* grad_input, float* grad_position, float* grad_weight);

/* ---- (2) stream + status variants ------------------------------------------------------------ */

int roitr_furthestsampling(int b, int n_max, const float* xyz, const int* offset, const int* new_offset, float* tmp,
                           int* idx, roitr_stream_t stream);
/* The sampling chain of a hierarchy (model/model.py:56-64 called level after level): the next level samples THIS level's picks in
 * pick order from the same first point, so while every arg-max among this level's first m / track_div picks was attained by
 * exactly one point, the next level's result is the prefix 0 .. m'-1 of those picks (csrc/pointops_fps.hip).  tie_out (b ints,
 * device; optional) receives per cloud the first pick index with a shared arg-max (INT_MAX: none in the tracked range);
 * prev_tie (optional) = the tie_out of the call whose picks this call samples: covered clouds are answered with the prefix
 * without running the chain, the others run it.  Results are bit-identical to roitr_furthestsampling either way. */
int roitr_furthestsampling_ex(int b, int n_max, const float* xyz, const int* offset, const int* new_offset, float* tmp, int* idx,
                              const int* prev_tie, int* tie_out, int track_div, roitr_stream_t stream);

/* Workspace for roitr_knn_build_grid / roitr_knnquery_ex: b clouds, n reference points, at most m queries. */
size_t roitr_knn_workspace_bytes(int b, int n, int m);

/* Counting-sorts the reference clouds into per-cloud uniform grids inside `ws`. */
int roitr_knn_build_grid(int b, int n, int m_capacity, const float* xyz, const int* offset, void* ws, roitr_stream_t stream);
/* same, with the mean points-per-cell the cell size is chosen for (<= 0: default 6; ~ (k+1)/3 for the largest k queried) */
int roitr_knn_build_grid_ex(int b, int n, int m_capacity, const float* xyz, const int* offset, void* ws, float target_occupancy,
                            roitr_stream_t stream);

/* The counting-sorted reference points inside a built workspace: float4 (x, y, z, original index as int bits), n entries,
 * cell-major order.  Valid after roitr_knn_build_grid on (b, n, m_capacity, ws). */
const void* roitr_knn_sorted_points(int b, int n, int m_capacity, void* ws);

/* Exact kNN.  Any of idx/dist2/group_idx/ppf may be NULL.
 *   group_idx (m, nsample-1): columns 1.. of idx  == pointops.queryandgroup(nsample-1, ..., return_idx=True)
 *                             (functions/pointops.py:88-89: kNN(k+1), drop column 0)
 *   ppf (m, nsample-1, 4):    lib/utils.py:358-389 calc_ppf_gpu(new_xyz, query_normals, xyz[group_idx],
 *                             ref_normals[group_idx])
 * use_grid != 0 needs a prior roitr_knn_build_grid(b, n, m_capacity, xyz, offset, ws). */
int roitr_knnquery_ex(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz, const int* offset,
                      const int* new_offset, int* idx, float* dist2, int* group_idx, float* ppf,
                      const float* ref_normals, const float* query_normals, int use_grid, int m_capacity, void* ws,
                      roitr_stream_t stream);

/* ------------------------------------------------------------------ input preparation
 * dataset/tdmatch.py:120-127: open3d estimate_normals(KDTreeSearchParamKNN(knn)) + dataset/common.py:312-320 normal_redirect.
 * normals (n,3) = unit eigenvector of the smallest eigenvalue of the covariance of the knn nearest points (the point
 * itself included), flipped towards view_point (3 host floats; NULL = leave Open3D's arbitrary sign).  use_grid as in
 * roitr_knnquery_ex (this call builds the grid itself).  ws: roitr_normals_workspace_bytes(b, n, knn). */
size_t roitr_normals_workspace_bytes(int b, int n, int knn);
int roitr_estimate_normals(int b, int n, const float* xyz, const int* offset, int knn, int use_grid, const float* view_point,
                           float* normals, void* ws, roitr_stream_t stream);
/* dataset/common.py:312-320 on its own: out = (dot(view_point - p, n) < 0) ? -n : n */
int roitr_normal_redirect(int n, const float* xyz, const float* normals_in, const float* view_point, float* normals_out,
                          roitr_stream_t stream);

/* kNN(1) distances for a radius test (lib/utils.py:509-521 get_node_occlusion_score): exact when < cap2, otherwise any
 * value >= cap2.  Same workspace / grid rules as roitr_knnquery_ex. */
int roitr_knn_within(int b, int n, int m, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset,
                     float cap2, float* dist2, int use_grid, int m_capacity, void* ws, roitr_stream_t stream);

int roitr_grouping_forward(int m, int nsample, int c, const float* input, const int* idx, float* output, roitr_stream_t stream);
int roitr_grouping_backward(int m, int nsample, int c, const float* grad_output, const int* idx, float* grad_input, roitr_stream_t stream);
int roitr_interpolation_forward(int n, int c, int k, const float* input, const int* idx, const float* weight, float* output, roitr_stream_t stream);
int roitr_interpolation_backward(int n, int c, int k, const float* grad_output, const int* idx, const float* weight, float* grad_input, roitr_stream_t stream);
int roitr_subtraction_forward(int n, int nsample, int c, const float* input1, const float* input2, const int* idx, float* output, roitr_stream_t stream);
int roitr_subtraction_backward(int n, int nsample, int c, const int* idx, const float* grad_output, float* grad_input1, float* grad_input2, roitr_stream_t stream);
int roitr_aggregation_forward(int n, int nsample, int c, int w_c, const float* input, const float* position, const float* weight, const int* idx, float* output, roitr_stream_t stream);
int roitr_aggregation_backward(int n, int nsample, int c, int w_c, const float* input, const float* position, const float* weight, const int* idx, const float* grad_output, float* grad_input, float* grad_position, float* grad_weight, roitr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ROITR_POINTOPS_H */
