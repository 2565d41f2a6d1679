// Ground-truth side outputs of the forward pass (need rot / trans): per-node occlusion scores
// (lib/utils.py:474-527 get_node_occlusion_score) and overlap-based node correspondences
// (lib/utils.py:530-614 get_node_correspondences), as called at model/RIGA_v2.py:91-111 (ref = tgt, src = src).
// Batched over pairs; clouds laid out [src_0..src_{B-1}, tgt_0..tgt_{B-1}].
#include "common.h"
#include "roitr_engine.h"
#include <algorithm>

namespace {

__device__ __forceinline__ float sq_norm3(float x, float y, float z)
{
#pragma clang fp contract(off)
    const float a = x * x, b = y * y, c = z * z;
    return (a + b) + c;
}
// lib/utils.py:139-156 for 3-vectors in torch-CPU arithmetic (see matching.hip)
__device__ __forceinline__ float square_distance3(float sx, float sy, float sz, float tx, float ty, float tz)
{
#pragma clang fp contract(off)
    const float m = sx * tx;
    const float xy = __fmaf_rn(sz, tz, __fmaf_rn(sy, ty, m));
    const float a = -2.0f * xy;
    const float b = a + sq_norm3(sx, sy, sz);
    return fmaxf(b + sq_norm3(tx, ty, tz), 1e-12f);
}
// p @ rot.T + trans  (torch.matmul over k = 3: fma chain in k order, then the add)
__device__ __forceinline__ void transform3(const float* __restrict__ R, const float* __restrict__ t, float x, float y, float z,
                                           float& ox, float& oy, float& oz)
{
#pragma clang fp contract(off)
    const float m0 = x * R[0], m1 = x * R[3], m2 = x * R[6];
    ox = __fmaf_rn(z, R[2], __fmaf_rn(y, R[1], m0)) + t[0];
    oy = __fmaf_rn(z, R[5], __fmaf_rn(y, R[4], m1)) + t[1];
    oz = __fmaf_rn(z, R[8], __fmaf_rn(y, R[7], m2)) + t[2];
}

// Padded clouds for the occlusion kNN(1): src_padded @ rot.T + trans (N_s + 1 rows: the zero pad row is transformed
// too, lib/utils.py:506) and tgt_padded (N_t + 1 rows).  One thread per output row over both halves.
__global__ void build_padded_kernel(int B, const float* __restrict__ pts, const int* __restrict__ pt_offset, const float* __restrict__ rot,
                                    const float* __restrict__ trans, float* __restrict__ out, int total_rows)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total_rows) return;
    // padded offsets: cloud c starts at pt_offset[c-1] + c
    const int c = segment_of(t, pt_offset, 2 * B, 1, 1);   // first c with t < pt_offset[c] + c + 1
    const int p0 = c == 0 ? 0 : pt_offset[c - 1];
    const int local = t - (p0 + c);
    const int n = pt_offset[c] - p0;
    float x = 0.f, y = 0.f, z = 0.f;
    if (local < n) { const float* p = pts + (size_t)(p0 + local) * 3; x = p[0]; y = p[1]; z = p[2]; }
    if (c < B) transform3(rot + (size_t)c * 9, trans + (size_t)c * 3, x, y, z, x, y, z);
    out[(size_t)t * 3] = x; out[(size_t)t * 3 + 1] = y; out[(size_t)t * 3 + 2] = z;
}

// padded offsets: off_pad[c] = pt_offset[c] + c + 1 for the 2B clouds, followed by B target offsets relative to the
// first target row (the target half used as a stand-alone query / reference set)
__global__ void padded_offsets_kernel(int NC, const int* __restrict__ pt_offset, int* __restrict__ off_pad)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int B = NC / 2;
    if (c < NC) off_pad[c] = pt_offset[c] + c + 1;
    if (c < B) off_pad[NC + c] = (pt_offset[B + c] + B + c + 1) - (pt_offset[B - 1] + B);
}

// score[node] = sum_k overlap[knn[node,k]] * mask / (sum mask + 1e-10) * node_mask    (lib/utils.py:511-527)
// overlap[p] = sqrt(d2[p]) < thr for the padded point p of the node's own cloud.
__global__ void occ_score_kernel(int n_nodes, int limit, const int* __restrict__ cloud_of_node, const int* __restrict__ pt_offset,
                                 const int* __restrict__ knn_idx, const int* __restrict__ knn_mask, const int* __restrict__ node_masks,
                                 const float* __restrict__ d2_pad, float thr, float* __restrict__ out)
{
    const int node = blockIdx.x * 256 + threadIdx.x;
    if (node >= n_nodes) return;
    const int c = cloud_of_node[node];
    const int base = (c == 0 ? 0 : pt_offset[c - 1]) + c;  // start of the cloud's padded rows
    float s = 0.f, m = 0.f;
    for (int k = 0; k < limit; ++k) {
        const int li = knn_idx[(size_t)node * limit + k];  // pad index n_c -> the pad row
        const float ov = sqrtf(d2_pad[base + li]) < thr ? 1.f : 0.f;
        const float mk = knn_mask[(size_t)node * limit + k] ? 1.f : 0.f;
        s += ov * mk; m += mk;
    }
    out[node] = s / (m + 1e-10f) * (node_masks[node] ? 1.f : 0.f);
}

// per node: its (transformed, for source nodes) position and the radius of its patch = max masked |p_k - node|
// (lib/utils.py:577-582), so that the pair kernel can apply the enclosing-sphere test before touching any point
__global__ __launch_bounds__(64) void node_radius_kernel(RoitrNodeCorr a, float* __restrict__ nodes_t, float* __restrict__ radius)
{
    const int node = blockIdx.x, k = threadIdx.x;
    const int B = a.pairs, L = a.limit;
    const int c = segment_of(node, a.node_offset, 2 * B);
    const int p0 = c == 0 ? 0 : a.pt_offset[c - 1], pn = a.pt_offset[c] - p0;
    float nx = a.nodes[(size_t)node * 3], ny = a.nodes[(size_t)node * 3 + 1], nz = a.nodes[(size_t)node * 3 + 2];
    const bool src = c < B;
    const float* R = a.rot + (size_t)(src ? c : 0) * 9; const float* T = a.trans + (size_t)(src ? c : 0) * 3;
    if (src) transform3(R, T, nx, ny, nz, nx, ny, nz);
    float d = 0.f;
    if (k < L && a.knn_mask[(size_t)node * L + k]) {
        const int li = a.knn_idx[(size_t)node * L + k];
        float x = 0.f, y = 0.f, z = 0.f;
        if (li < pn) { const float* p = a.points + (size_t)(p0 + li) * 3; x = p[0]; y = p[1]; z = p[2]; }
        if (src) transform3(R, T, x, y, z, x, y, z);
        const float dx = x - nx, dy = y - ny, dz = z - nz;
        d = sqrtf(dx * dx + dy * dy + dz * dz);
    }
    d = wave_max(d);
    if (k == 0) { nodes_t[(size_t)node * 3] = nx; nodes_t[(size_t)node * 3 + 1] = ny; nodes_t[(size_t)node * 3 + 2] = nz; radius[node] = d; }
}

// Candidate (pair, ref node i, src node j) triples: one THREAD per triple evaluates the node masks and the enclosing-sphere prune
// (l.577-586) from the per-node records, writes the 0 of a pruned / masked entry and appends the survivors to a work list --
// the patch-overlap kernel below then only runs on those (one workgroup per triple used to cost 2.1 ms per 512-pair step for
// 3.1 M workgroups of which most left after two scalar loads).  The list lives in out_idx (free until the final compaction), its
// counter in out_count[0]; an entry is the flat index pair * max_nodes^2 + i * max_nodes + j.  Every entry of `overlap` is
// written at its own position, so the order of the list does not matter.
__global__ __launch_bounds__(256) void node_corr_prune_kernel(RoitrNodeCorr a)
{
    __shared__ int wcnt[4], wbase[4];
    const int pair = blockIdx.y;
    const int mm = a.max_nodes * a.max_nodes;
    const int rem = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int B = a.pairs, sc = pair, tc = B + pair;
    const int s0 = sc == 0 ? 0 : a.node_offset[sc - 1], ns = a.node_offset[sc] - s0;
    const int t0 = a.node_offset[tc - 1], nt = a.node_offset[tc] - t0;
    bool keep = false;
    if (rem < mm) {
        const int i = rem / a.max_nodes, j = rem - i * a.max_nodes;
        if (i < nt && j < ns) {
            const int rnode = t0 + i, snode = s0 + j;
            bool live = a.node_masks[rnode] && a.node_masks[snode];
            if (live) {
                const float* rn = a.nodes_t + (size_t)rnode * 3; const float* sn_ = a.nodes_t + (size_t)snode * 3;
                const float nd0 = sqrtf(square_distance3(rn[0], rn[1], rn[2], sn_[0], sn_[1], sn_[2]));
                live = a.radius[rnode] + a.radius[snode] + a.pos_radius - nd0 > 0.f;
            }
            if (live) keep = true;
            else a.overlap[(size_t)pair * a.mat_stride + (size_t)i * ns + j] = 0.f;
        }
    }
    // one atomic per workgroup: wave counts -> block base -> lane rank
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        int base = tot ? atomicAdd(a.out_count, tot) : 0;
        for (int w = 0; w < 4; ++w) { wbase[w] = base; base += wcnt[w]; }
    }
    __syncthreads();
    if (keep) a.out_idx[wbase[wave] + __popcll(m & ((1ull << lane) - 1ull))] = pair * mm + rem;
}

// a workgroup per surviving (pair, ref node i, src node j): overlap ratio of the two patches
__global__ __launch_bounds__(256) void node_corr_kernel(RoitrNodeCorr a)
{
    __shared__ float rp[64][3], sp[64][3];
    __shared__ int rm[64], sm[64];
    __shared__ int rhit[64], shit[64];
    __shared__ float rmax_s, smax_s;
    const int B = a.pairs, L = a.limit;
    const int tid = threadIdx.x;
    const int n_work = a.out_count[0];
    const int mm = a.max_nodes * a.max_nodes;
    for (int wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
    const int code = a.out_idx[wi];
    const int pair = code / mm, rem = code - pair * mm;
    const int i = rem / a.max_nodes, j = rem - i * a.max_nodes;
    const int sc = pair, tc = B + pair;
    const int s0 = sc == 0 ? 0 : a.node_offset[sc - 1], ns = a.node_offset[sc] - s0;
    const int t0 = a.node_offset[tc - 1];
    float* outm = a.overlap + (size_t)pair * a.mat_stride;
    const int rnode = t0 + i, snode = s0 + j;
    const float* R = a.rot + (size_t)pair * 9; const float* T = a.trans + (size_t)pair * 3;
    __syncthreads();   // the previous triple's shared records are no longer read
    const int tp0 = a.pt_offset[tc - 1], tn = a.pt_offset[tc] - tp0;
    const int sp0 = sc == 0 ? 0 : a.pt_offset[sc - 1], sn = a.pt_offset[sc] - sp0;
    const float rnx = a.nodes[(size_t)rnode * 3], rny = a.nodes[(size_t)rnode * 3 + 1], rnz = a.nodes[(size_t)rnode * 3 + 2];
    float snx, sny, snz;
    transform3(R, T, a.nodes[(size_t)snode * 3], a.nodes[(size_t)snode * 3 + 1], a.nodes[(size_t)snode * 3 + 2], snx, sny, snz);
    if (tid < 64) {
        const int k = tid;
        const int li = a.knn_idx[(size_t)rnode * L + k];
        float x = 0.f, y = 0.f, z = 0.f;
        if (li < tn) { const float* p = a.points + (size_t)(tp0 + li) * 3; x = p[0]; y = p[1]; z = p[2]; }
        rp[k][0] = x; rp[k][1] = y; rp[k][2] = z; rm[k] = a.knn_mask[(size_t)rnode * L + k]; rhit[k] = 0;
    } else if (tid < 128) {
        const int k = tid - 64;
        const int li = a.knn_idx[(size_t)snode * L + k];
        float x = 0.f, y = 0.f, z = 0.f;
        if (li < sn) { const float* p = a.points + (size_t)(sp0 + li) * 3; x = p[0]; y = p[1]; z = p[2]; }
        transform3(R, T, x, y, z, x, y, z);
        sp[k][0] = x; sp[k][1] = y; sp[k][2] = z; sm[k] = a.knn_mask[(size_t)snode * L + k]; shit[k] = 0;
    }
    __syncthreads();
    // enclosing-sphere prune (l.577-586)
    if (tid < 64) {
        float d = 0.f;
        if (rm[tid]) { const float dx = rp[tid][0] - rnx, dy = rp[tid][1] - rny, dz = rp[tid][2] - rnz; d = sqrtf(dx * dx + dy * dy + dz * dz); }
        d = wave_max(d);
        if (tid == 0) rmax_s = d;
    } else if (tid < 128) {
        const int k = tid - 64;
        float d = 0.f;
        if (sm[k]) { const float dx = sp[k][0] - snx, dy = sp[k][1] - sny, dz = sp[k][2] - snz; d = sqrtf(dx * dx + dy * dy + dz * dz); }
        d = wave_max(d);
        if (k == 0) smax_s = d;
    }
    __syncthreads();
    const float nd = sqrtf(square_distance3(rnx, rny, rnz, snx, sny, snz));
    if (!(rmax_s + smax_s + a.pos_radius - nd > 0.f)) { if (tid == 0) outm[(size_t)i * ns + j] = 0.f; continue; }   // block-uniform
    const float r2 = a.pos_radius * a.pos_radius;
    for (int e = tid; e < 64 * 64; e += 256) {
        const int p = e >> 6, q = e & 63;
        if (rm[p] && sm[q]) {
            const float d = square_distance3(rp[p][0], rp[p][1], rp[p][2], sp[q][0], sp[q][1], sp[q][2]);
            if (d < r2) { rhit[p] = 1; shit[q] = 1; }
        }
    }
    __syncthreads();
    if (tid < 64) {
        float rc = (float)rhit[tid], sc_ = (float)shit[tid], rmk = rm[tid] ? 1.f : 0.f, smk = sm[tid] ? 1.f : 0.f;
        rc = wave_sum(rc); sc_ = wave_sum(sc_); rmk = wave_sum(rmk); smk = wave_sum(smk);
        if (tid == 0) outm[(size_t)i * ns + j] = (rc / rmk + sc_ / smk) / 2.0f;
    }    }
}

// row-major compaction of the positive entries of each pair's (n_t, n_s) overlap matrix (torch.nonzero order, l.605-612)
__global__ __launch_bounds__(1024) void node_corr_compact_kernel(RoitrNodeCorr a)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int pair = blockIdx.x, B = a.pairs;
    const int sc = pair, tc = B + pair;
    const int ns = a.node_offset[sc] - (sc == 0 ? 0 : a.node_offset[sc - 1]);
    const int nt = a.node_offset[tc] - a.node_offset[tc - 1];
    const float* m = a.overlap + (size_t)pair * a.mat_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    const int total = nt * ns;
    for (int base = 0; base < total; base += 1024) {
        const int e = base + tid;
        const float v = e < total ? m[e] : 0.f;
        const int f = v > 0.f ? 1 : 0;
        int incl = f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wb = 0;
        for (int w = 0; w < wave; ++w) wb += wsum[w];
        const int carry = carry_s;
        if (f) {
            const int pos = carry + wb + incl - 1;
            a.out_idx[((size_t)pair * a.mat_stride + pos) * 2] = e / ns;
            a.out_idx[((size_t)pair * a.mat_stride + pos) * 2 + 1] = e % ns;
            a.out_overlap[(size_t)pair * a.mat_stride + pos] = v;
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + wb + incl;
        __syncthreads();
    }
    if (tid == 0) a.out_count[pair] = carry_s;
}

}  // namespace

extern "C" int roitr_build_padded_clouds(int pairs, int n_points, const float* pts, const int* pt_offset, const float* rot,
                                         const float* trans, float* out_pts, int* out_offset, hipStream_t stream)
{
    const int NC = 2 * pairs, total = n_points + NC;
    padded_offsets_kernel<<<div_up(NC, 256), 256, 0, stream>>>(NC, pt_offset, out_offset);
    ROITR_LAUNCH_CHECK();
    build_padded_kernel<<<div_up(total, 256), 256, 0, stream>>>(pairs, pts, pt_offset, rot, trans, out_pts, total);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_node_occlusion_score(int n_nodes, int limit, const int* cloud_of_node, const int* pt_offset, const int* knn_idx,
                                          const int* knn_mask, const int* node_masks, const float* d2_padded, float overlap_thres,
                                          float* out, hipStream_t stream)
{
    if (n_nodes <= 0) return ROITR_OK;
    occ_score_kernel<<<div_up(n_nodes, 256), 256, 0, stream>>>(n_nodes, limit, cloud_of_node, pt_offset, knn_idx, knn_mask, node_masks,
                                                                d2_padded, overlap_thres, out);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_node_correspondences(const RoitrNodeCorr* a, hipStream_t stream)
{
    if (a->pairs <= 0) return ROITR_OK;
    if (a->limit != 64) return ROITR_ERR_UNSUPPORTED;
    if (!a->nodes_t || !a->radius) return ROITR_ERR_ARG;
    node_radius_kernel<<<a->n_nodes, 64, 0, stream>>>(*a, a->nodes_t, a->radius);
    ROITR_LAUNCH_CHECK();
    const long total = (long)a->pairs * a->max_nodes * a->max_nodes;
    if (total >= 0x7fffffffL || a->mat_stride < (long)a->max_nodes * a->max_nodes) return ROITR_ERR_UNSUPPORTED;
    ROITR_HIP(hipMemsetAsync(a->out_count, 0, sizeof(int), stream));   // the work-list counter (overwritten by the compaction)
    node_corr_prune_kernel<<<dim3((unsigned)((a->max_nodes * a->max_nodes + 255) / 256), (unsigned)a->pairs), 256, 0, stream>>>(*a);
    ROITR_LAUNCH_CHECK();
    node_corr_kernel<<<(unsigned)std::min<long>(total, 16384), 256, 0, stream>>>(*a);
    ROITR_LAUNCH_CHECK();
    node_corr_compact_kernel<<<a->pairs, 1024, 0, stream>>>(*a);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}
