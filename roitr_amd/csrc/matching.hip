// Coarse-to-fine matching tail of the RoITr forward pass (reference model/RIGA_v2.py:82-173):
// point-to-node partition, coarse (superpoint) matching, patch assembly, optimal transport,
// fine (point) matching with deterministic row-major compaction.
//
// All kernels are batched over `pairs`: clouds are laid out [src_0..src_{B-1}, tgt_0..tgt_{B-1}].
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include "prof.h"
#include "roitr_engine.h"

namespace {

// hipcc's __fmul_rn/__fadd_rn are plain operators (re-fusable), so contraction is switched off explicitly
__device__ __forceinline__ float sq_norm3(float x, float y, float z)
{
#pragma clang fp contract(off)
    const float a = x * x, b = y * y, c = z * z;
    return (a + b) + c;
}
// lib/utils.py:139-156 square_distance for 3-vectors in torch-CPU arithmetic:
// dist = -2*matmul (fma chain over k) ; dist += |src|^2 ; dist += |tgt|^2 ; clamp(min=1e-12)
__device__ __forceinline__ float square_distance3(float sx, float sy, float sz, float s2, float tx, float ty, float tz, float t2)
{
#pragma clang fp contract(off)
    const float m = sx * tx;
    const float xy = __fmaf_rn(sz, tz, __fmaf_rn(sy, ty, m));
    const float a = -2.0f * xy;
    const float b = a + s2;
    return fmaxf(b + t2, 1e-12f);
}

// ------------------------------------------------------------------ point_to_node_partition (lib/utils.py:428-471)
// A: every point -> nearest node of its cloud (first minimum), distance kept for step B
__global__ __launch_bounds__(256) void p2n_assign_kernel(int n_points, const float* __restrict__ pts, const int* __restrict__ pt_offset,
                                                         const int* __restrict__ cloud_of_pt_hint, const float* __restrict__ nodes,
                                                         const int* __restrict__ node_offset, int b, int* __restrict__ p2n,
                                                         float* __restrict__ p2n_dist, int* __restrict__ node_masks)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_points) return;
    const int c = segment_of(i, pt_offset, b);
    (void)cloud_of_pt_hint;
    const int ns = c == 0 ? 0 : node_offset[c - 1], ne = node_offset[c];
    const float px = pts[(size_t)i * 3], py = pts[(size_t)i * 3 + 1], pz = pts[(size_t)i * 3 + 2];
    const float p2 = sq_norm3(px, py, pz);
    float best = INFINITY; int bi = ns;
    for (int j = ns; j < ne; ++j) {
        const float nx = nodes[(size_t)j * 3], ny = nodes[(size_t)j * 3 + 1], nz = nodes[(size_t)j * 3 + 2];
        const float d = square_distance3(nx, ny, nz, sq_norm3(nx, ny, nz), px, py, pz, p2);
        if (d < best) { best = d; bi = j; }
    }
    p2n[i] = bi - ns;  // node index local to the cloud, as the reference returns it
    p2n_dist[i] = best;
    node_masks[bi] = 1;
}

// B: one block per node: the `limit` nearest OWNED points, ascending (distance, index); pad = n_c (cloud size).
// Owned points are compacted into LDS in one pass over the cloud and bitonic-sorted there (a node owns
// ~N/n = 64 points on average); a node owning more than P2N_CAP points falls back to repeated arg-min.
constexpr int P2N_CAP = 2048;
__global__ __launch_bounds__(256) void p2n_topk_kernel(const int* __restrict__ pt_offset, const int* __restrict__ node_offset,
                                                       const int* __restrict__ cloud_of_node, const int* __restrict__ p2n,
                                                       const float* __restrict__ p2n_dist, int limit, int* __restrict__ knn_idx,
                                                       int* __restrict__ knn_mask)
{
    __shared__ unsigned long long keys[P2N_CAP];
    __shared__ unsigned long long red[4];
    __shared__ unsigned long long last_s;
    __shared__ int cnt_s;
    const int node = blockIdx.x;
    const int c = cloud_of_node[node];
    const int ps = c == 0 ? 0 : pt_offset[c - 1], pe = pt_offset[c];
    const int local = node - (c == 0 ? 0 : node_offset[c - 1]);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) cnt_s = 0;
    __syncthreads();
    for (int j = ps + tid; j < pe; j += 256) {
        if (p2n[j] == local) {
            const int pos = atomicAdd(&cnt_s, 1);
            if (pos < P2N_CAP) keys[pos] = ((unsigned long long)__float_as_uint(p2n_dist[j]) << 32) | (unsigned)(j - ps);
        }
    }
    __syncthreads();
    const int count = cnt_s;
    if (count <= P2N_CAP) {
        int cap = 64;
        while (cap < count) cap <<= 1;
        for (int e = count + tid; e < cap; e += 256) keys[e] = ~0ull;
        __syncthreads();
        for (int k = 2; k <= cap; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int e = tid; e < cap; e += 256) {
                    const int q = e ^ j;
                    if (q > e) {
                        const unsigned long long x = keys[e], y = keys[q];
                        const bool up = (e & k) == 0;
                        if ((x > y) == up) { keys[e] = y; keys[q] = x; }
                    }
                }
                __syncthreads();
            }
        }
        for (int t = tid; t < limit; t += 256) {
            const bool ok = t < count;
            knn_idx[(size_t)node * limit + t] = ok ? (int)(unsigned)keys[t] : (pe - ps);
            knn_mask[(size_t)node * limit + t] = ok ? 1 : 0;
        }
        return;
    }
    // ---- rare: more owned points than the LDS buffer holds
    unsigned long long last = 0ull;
    bool first = true;
    for (int t = 0; t < limit; ++t) {
        unsigned long long best = ~0ull;
        for (int j = ps + tid; j < pe; j += 256) {
            if (p2n[j] == local) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(p2n_dist[j]) << 32) | (unsigned)(j - ps);
                if ((first || key > last) && key < best) best = key;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long w = __shfl_xor(best, o, 64); best = w < best ? w : best; }
        if (lane == 0) red[wave] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long bb = red[0];
            for (int w = 1; w < 4; ++w) bb = red[w] < bb ? red[w] : bb;
            last_s = bb;
            knn_idx[(size_t)node * limit + t] = (int)(unsigned)bb;
            knn_mask[(size_t)node * limit + t] = 1;
        }
        __syncthreads();
        last = last_s; first = false;
    }
}

constexpr int COARSE_LDS_KEYS = 16384;   // 128 KB of 64-bit keys
// ------------------------------------------------------------------ block-wide "the `want` smallest 64-bit keys, sorted"
// keys: LDS, lds_cap (power of two) entries.  total <= lds_cap: one bitonic sort.  Larger (e.g. 468 x 468 superpoint pairs
// of a 30000-point cloud): the keys are sorted in LDS-sized chunks, every chunk contributes its W = pow2(want) smallest to
// `spill` (global), and the spilled winners are reduced the same way until they fit -- the result is exactly the
// `want` smallest keys of the whole set, ascending, in keys[0..want).
__device__ __forceinline__ void block_bitonic_sort(unsigned long long* keys, int cap, int tid)
{
    for (int k = 2; k <= cap; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int e = tid; e < cap; e += 1024) {
                const int p = e ^ j;
                if (p > e) {
                    const unsigned long long x = keys[e], y = keys[p];
                    const bool up = (e & k) == 0;
                    if ((x > y) == up) { keys[e] = y; keys[p] = x; }
                }
            }
            __syncthreads();
        }
    }
}

template <typename KeyFn>
__device__ __forceinline__ void block_select_smallest(KeyFn key_of, int total, int want, unsigned long long* keys, int lds_cap,
                                                      unsigned long long* spill, int tid)
{
    int W = 1;
    while (W < want) W <<= 1;
    int count = total;
    bool from_spill = false;
    while (true) {
        if (count <= lds_cap) {
            int cap = 1;
            while (cap < count) cap <<= 1;
            for (int e = tid; e < cap; e += 1024) keys[e] = e < count ? (from_spill ? spill[e] : key_of(e)) : ~0ull;
            __syncthreads();
            block_bitonic_sort(keys, cap, tid);
            return;
        }
        const int nchunks = (count + lds_cap - 1) / lds_cap;
        for (int c = 0; c < nchunks; ++c) {
            const int base = c * lds_cap;
            for (int e = tid; e < lds_cap; e += 1024) {
                const int g = base + e;
                keys[e] = g < count ? (from_spill ? spill[g] : key_of(g)) : ~0ull;
            }
            __syncthreads();
            block_bitonic_sort(keys, lds_cap, tid);
            // chunk c's winners land at spill[c W ..): behind everything this and later chunks still read (W <= lds_cap)
            for (int e = tid; e < W; e += 1024) spill[(size_t)c * W + e] = keys[e];
            __syncthreads();
        }
        __threadfence_block();
        count = nchunks * W;
        from_spill = true;
    }
}

// ------------------------------------------------------------------ CoarseMatching (model/modules.py:141-178)
// one block (1024 threads) per pair.  ref = tgt nodes, src = src nodes (RIGA_v2.py:121).
// scores[i][j] = exp(-square_distance(ref_i, src_j)) over mask-valid rows/cols, dual normalisation,
// top-`num` by (score desc, flat index asc).  Scratch: n_r*n_s floats per pair.
__global__ __launch_bounds__(1024) void coarse_match_kernel(RoitrCoarse a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int pair = blockIdx.x, B = a.pairs;
    const int sc = pair, tc = B + pair;  // cloud ids
    const int s0 = sc == 0 ? 0 : a.node_offset[sc - 1], s1 = a.node_offset[sc];
    const int t0 = a.node_offset[tc - 1], t1 = a.node_offset[tc];
    const int nr = t1 - t0, nsr = s1 - s0;  // ref rows (tgt), src cols
    const int tid = threadIdx.x;
    float* M = a.scratch + (size_t)pair * a.scratch_stride;
    float* rowsum = M + (size_t)nr * nsr;
    float* colsum = rowsum + nr;
    const int C = a.C;
    // squared norms with a plain sequential sum (tolerance-level parity: torch.sum is a pairwise tree)
    for (int i = tid; i < nr + nsr; i += 1024) {
        const float* f = a.feats + (size_t)(i < nr ? t0 + i : s0 + (i - nr)) * C;
        float s = 0.f;
        for (int k = 0; k < C; ++k) s += f[k] * f[k];
        (i < nr ? rowsum : colsum)[i < nr ? i : i - nr] = s;  // temporarily: norms
    }
    __syncthreads();
    const float* XY = a.xy ? a.xy + (size_t)pair * a.xy_stride : nullptr;
    for (int e = tid; e < nr * nsr; e += 1024) {
        const int i = e / nsr, j = e % nsr;
        const bool ok = a.node_masks[t0 + i] && a.node_masks[s0 + j];
        float v = -1.f;
        if (ok) {
            float xy = 0.f;
            if (XY) xy = XY[(size_t)i * a.xy_ld + j];  // k-ordered fma chain from the MFMA GEMM: same value as below
            else {
                const float* fr = a.feats + (size_t)(t0 + i) * C;
                const float* fs = a.feats + (size_t)(s0 + j) * C;
                for (int k = 0; k < C; ++k) xy = __fmaf_rn(fr[k], fs[k], xy);
            }
            const float d = fmaxf((-2.0f * xy + rowsum[i]) + colsum[j], 1e-12f);
            v = expf(-d);
        }
        M[e] = v;
    }
    __syncthreads();
    // row / column sums over valid entries (sequential order within a row/col); the norms are dead from here on
    for (int i = tid; i < nr + nsr; i += 1024) {
        float s = 0.f;
        if (i < nr) { for (int j = 0; j < nsr; ++j) { const float v = M[(size_t)i * nsr + j]; if (v >= 0.f) s += v; } rowsum[i] = s; }
        else { const int j = i - nr; for (int r = 0; r < nr; ++r) { const float v = M[(size_t)r * nsr + j]; if (v >= 0.f) s += v; } colsum[j] = s; }
    }
    __syncthreads();
    int nvr = 0, nvs = 0;
    for (int i = 0; i < nr; ++i) nvr += a.node_masks[t0 + i] ? 1 : 0;
    for (int j = 0; j < nsr; ++j) nvs += a.node_masks[s0 + j] ? 1 : 0;
    const int num = min(a.num_corr, nvr * nvs);
    // keys: (~score bits, flat index) ascending == score descending, reference flat order
    const int total = nr * nsr;
    auto key_of = [&](int e) -> unsigned long long {
        const float v = M[e];
        if (!(v >= 0.f)) return ~0ull;
        const int i = e / nsr, j = e % nsr;
        float sv = v;
        if (a.dual_norm) sv = (v / (rowsum[i] + 1e-8f)) * (v / (colsum[j] + 1e-8f));
        return ((unsigned long long)(~__float_as_uint(sv)) << 32) | (unsigned)e;
    };
    block_select_smallest(key_of, total, max(num, 1), keys, a.lds_cap, reinterpret_cast<unsigned long long*>(((uintptr_t)(colsum + nsr + 8) + 7) & ~(uintptr_t)7), tid);
    for (int t = tid; t < a.num_corr; t += 1024) {
        int ri = -1, si = -1; float sv = 0.f;
        if (t < num) {
            const unsigned long long key = keys[t];
            const int e = (int)(unsigned)key;
            ri = e / nsr; si = e % nsr;
            sv = __uint_as_float(~(unsigned)(key >> 32));
        }
        a.tgt_corr[(size_t)pair * a.num_corr + t] = ri;
        a.src_corr[(size_t)pair * a.num_corr + t] = si;
        a.corr_scores[(size_t)pair * a.num_corr + t] = sv;
    }
    if (tid == 0) a.n_corr[pair] = num;
}

// ------------------------------------------------------------------ AdaptiveSuperPointMatching (model/modules.py:75-124)
// 4DMatch coarse matching, called like CoarseMatching (first argument = tgt side, RIGA_v2.py:121).
// dist = sqrt(clamp(2 - 2 xy, 1e-12)) over mask-valid pairs; if fewer than min(num, #valid) pairs satisfy
// dist <= threshold: the `min` smallest distances (ascending, flat index on ties), else ALL pairs under the threshold in
// row-major order.  scores = exp(-dist).  Output capacity per pair: num_corr slots (= n_t * n_s in the engine).
__global__ __launch_bounds__(1024) void adaptive_match_kernel(RoitrCoarse a, int min_num, float threshold)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    __shared__ int wsum[16];
    __shared__ int carry_s, below_s;
    const int pair = blockIdx.x, B = a.pairs;
    const int sc = pair, tc = B + pair;
    const int s0 = sc == 0 ? 0 : a.node_offset[sc - 1], nsr = a.node_offset[sc] - s0;
    const int t0 = a.node_offset[tc - 1], nr = a.node_offset[tc] - t0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* M = a.scratch + (size_t)pair * a.scratch_stride;
    const float* XY = a.xy + (size_t)pair * a.xy_stride;
    const int total = nr * nsr;
    if (tid == 0) { carry_s = 0; below_s = 0; }
    __syncthreads();
    int nvalid_local = 0, below_local = 0;
    for (int e = tid; e < total; e += 1024) {
        const int i = e / nsr, j = e % nsr;
        float v = -1.f;
        if (a.node_masks[t0 + i] && a.node_masks[s0 + j]) {
            v = sqrtf(fmaxf(2.0f - 2.0f * XY[(size_t)i * a.xy_ld + j], 1e-12f));
            nvalid_local++;
            below_local += v <= threshold ? 1 : 0;
        }
        M[e] = v;
    }
    atomicAdd(&below_s, below_local);
    atomicAdd(&carry_s, nvalid_local);
    __syncthreads();
    const int below = below_s, nvalid = carry_s;
    __syncthreads();
    const int kmin = min(min_num, nvalid);
    if (below < kmin) {
        // top-kmin smallest distances
        auto key_of = [&](int e) -> unsigned long long {
            return M[e] >= 0.f ? (((unsigned long long)__float_as_uint(M[e]) << 32) | (unsigned)e) : ~0ull;
        };
        block_select_smallest(key_of, total, max(kmin, 1), keys, a.lds_cap, reinterpret_cast<unsigned long long*>(((uintptr_t)(M + total + nr + nsr + 8) + 7) & ~(uintptr_t)7), tid);
        for (int t = tid; t < kmin; t += 1024) {
            const unsigned long long key = keys[t];
            const int e = (int)(unsigned)key;
            a.tgt_corr[(size_t)pair * a.num_corr + t] = e / nsr;
            a.src_corr[(size_t)pair * a.num_corr + t] = e % nsr;
            a.corr_scores[(size_t)pair * a.num_corr + t] = expf(-__uint_as_float((unsigned)(key >> 32)));
        }
        if (tid == 0) a.n_corr[pair] = kmin;
        return;
    }
    // all pairs under the threshold, row-major (torch.nonzero order)
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < total; base += 1024) {
        const int e = base + tid;
        const float v = e < total ? M[e] : -1.f;
        const int f = (v >= 0.f && v <= threshold) ? 1 : 0;
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
            if (pos < a.num_corr) {
                a.tgt_corr[(size_t)pair * a.num_corr + pos] = e / nsr;
                a.src_corr[(size_t)pair * a.num_corr + pos] = e % nsr;
                a.corr_scores[(size_t)pair * a.num_corr + pos] = expf(-v);
            }
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + wb + incl;
        __syncthreads();
    }
    if (tid == 0) a.n_corr[pair] = min(carry_s, a.num_corr);
}

// ------------------------------------------------------------------ patch assembly (RIGA_v2.py:125-147)
// per (pair, corr p, slot i): global feature-row index (or -1 = zero pad), knn point, mask, for both sides
// Patch slot -> (pair, position in the pair's coarse list).  Strided layout (pair_off == NULL): slot = pair * num_corr + p, live
// while p < n_corr[pair].  Compacted layout (round 6; the adaptive 4DMatch matching selects anything between 128 and n_t * n_s
// node pairs per cloud pair, RIGA_v2.py:126-152 runs the tail on the SELECTED ones only): the live patches of all pairs back to
// back, pair b owns slots [pair_off[b], pair_off[b + 1]); slots from pair_off[pairs] on are dead and never touched.
struct PatchSlot { int pair, p; bool live; };
__device__ __forceinline__ PatchSlot patch_slot(int slot, int pairs, int num_corr, const int* __restrict__ n_corr, const int* __restrict__ pair_off)
{
    PatchSlot r;
    if (pair_off) {
        r.live = slot < pair_off[pairs];
        r.pair = r.live ? segment_of(slot, pair_off + 1, pairs) : 0;
        r.p = slot - pair_off[r.pair];
    } else {
        r.pair = slot / num_corr; r.p = slot % num_corr;
        r.live = r.p < n_corr[r.pair];
    }
    return r;
}

// pair_off[0] = 0, pair_off[b + 1] = min(slots, n_corr[0] + ... + n_corr[b]): one block, pairs <= a few thousand
__global__ __launch_bounds__(1024) void patch_offsets_kernel(int pairs, const int* __restrict__ n_corr, int slots, int* __restrict__ pair_off)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { carry_s = 0; pair_off[0] = 0; }
    __syncthreads();
    for (int base = 0; base < pairs; base += 1024) {
        const int i = base + tid;
        int incl = i < pairs ? n_corr[i] : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wb = 0;
        for (int w = 0; w < wave; ++w) wb += wsum[w];
        const int carry = carry_s;
        if (i < pairs) pair_off[i + 1] = min(slots, carry + wb + incl);
        __syncthreads();
        if (tid == 1023) carry_s = carry + wb + incl;
        __syncthreads();
    }
}

__global__ void patch_gather_kernel(RoitrPatch a, long total)
{
    const long t = blockIdx.x * 256L + threadIdx.x;
    if (t >= total) return;
    const int slot = (int)(t / a.limit), i = (int)(t % a.limit);
    const PatchSlot ps = patch_slot(slot, a.pairs, a.num_corr, a.n_corr, a.pair_off);
    if (a.pair_off && !ps.live) return;   // compacted: dead slots are never read
    const int pair = ps.pair, p = ps.p;
    const bool live = ps.live;
#pragma unroll
    for (int side = 0; side < 2; ++side) {  // 0 = tgt (rows of the score matrix), 1 = src (columns)
        const int cloud = side == 0 ? a.pairs + pair : pair;
        const int n0 = cloud == 0 ? 0 : a.node_offset[cloud - 1];
        const int p0 = cloud == 0 ? 0 : a.pt_offset[cloud - 1], p1 = a.pt_offset[cloud];
        const int* corr = side == 0 ? a.tgt_corr : a.src_corr;
        int row = -1, mask = 0; float x = 0.f, y = 0.f, z = 0.f;
        if (live) {
            const int node = n0 + corr[(size_t)pair * a.num_corr + p];
            const int li = a.knn_idx[(size_t)node * a.limit + i];
            mask = a.knn_mask[(size_t)node * a.limit + i];
            if (li < p1 - p0) {  // index n_c selects the zero pad row (RIGA_v2.py:86-89)
                row = p0 + li;
                const float* q = a.points + (size_t)row * 3;
                x = q[0]; y = q[1]; z = q[2];
            }
        }
        const size_t o = (size_t)t;
        (side == 0 ? a.tgt_rows : a.src_rows)[o] = row;
        (side == 0 ? a.tgt_masks : a.src_masks)[o] = mask;
        float* pp = (side == 0 ? a.tgt_pts : a.src_pts) + o * 3;
        pp[0] = x; pp[1] = y; pp[2] = z;
    }
}

// ------------------------------------------------------------------ LearnableLogOptimalTransport (modules.py:10-72)
// One WAVE per patch; rows/cols = limit+1 = 65 (64 points + dustbin).  The log-domain Sinkhorn of the reference
//   u = log_mu - logsumexp_j(S + v),  v = log_nu - logsumexp_i(S + u)      (100 iterations)
// is run in the exponential domain with per-row max shifts: K'_ij = exp(S_ij - m_i), b = e^v,
// a~_i = mu_i / (K' b)_i (= e^{u_i + m_i}), b_j = nu_j / (K'^T a~)_j -- the same iteration, no exp/log inside the loop.
// Lane l keeps row l AND column l of the 64x64 block of K' in registers (128 VGPRs); the dustbin row/column are one
// value per lane.  An iteration is two 64-term dot products per lane against a vector that lives one element per
// lane: its elements reach the FMAs through DPP row rotations (a source modifier of the FMA) and three ds_bpermute
// row-block shifts -- no LDS reads, no barrier, no broadcast traffic (the first version read the vector back out of LDS
// with 32 ds_read_b128 per iteration and was bound by LDS bandwidth: 1.9 ms per 128-pair forward).
// Masked rows/cols carry mu = 0 / K' = 0 exactly (the reference's -1e6 entries underflow to 0 in its logsumexp too).
// The loop stops early only when b reproduced itself bit for bit (then every later iterate is identical).
constexpr int OTN = 65;
constexpr float OT_FAST_SPREAD = 30.f;   // widest score range of a row (dustbin included) the exponential form takes
// ACC += KV * (B of lane (row, (col - N) & 15)): the DPP row rotate rides as a source modifier of the FMA (hipcc emits a
// separate v_mov_b32_dpp per term from the builtin, which doubles the VALU work of the loop).  B is always written
// several instructions before its first use here (the three ds_bpermute sit in between), which covers the
// VALU-write -> DPP-read hazard the assembler cannot see inside inline asm.
#define OT_FMAC_DPP(ACC, B, KV, N_) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:" #N_ " row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(B), "v"(KV))
// s0..s3 += sum_n ror16<n>(B) * KARR[16 T + n]: one 16-lane row block of a 64-term dot product whose vector operand is
// distributed over the lanes (lane j holds element j)
#define OT_TERM(ACC, B, KARR, T_, N_) OT_FMAC_DPP(ACC, B, KARR[(T_) * 16 + (N_)], N_)
#define OT_BLOCK(B, KARR, T_)                                                                                              \
    s0 = fmaf(B, KARR[(T_) * 16], s0); OT_TERM(s1, B, KARR, T_, 1); OT_TERM(s2, B, KARR, T_, 2); OT_TERM(s3, B, KARR, T_, 3);     \
    OT_TERM(s0, B, KARR, T_, 4); OT_TERM(s1, B, KARR, T_, 5); OT_TERM(s2, B, KARR, T_, 6); OT_TERM(s3, B, KARR, T_, 7);     \
    OT_TERM(s0, B, KARR, T_, 8); OT_TERM(s1, B, KARR, T_, 9); OT_TERM(s2, B, KARR, T_, 10); OT_TERM(s3, B, KARR, T_, 11);   \
    OT_TERM(s0, B, KARR, T_, 12); OT_TERM(s1, B, KARR, T_, 13); OT_TERM(s2, B, KARR, T_, 14); OT_TERM(s3, B, KARR, T_, 15)
// K . x for the lane-distributed 64-vector x: the three other row blocks of x arrive by ds_bpermute (LDS crossbar, no
// LDS storage), every product is then one DPP-modified FMA -- no LDS reads in the Sinkhorn loop at all
__device__ __forceinline__ float ot_dot64(const float (&KARR)[64], float x, float init, int lane)
{
    const float x1 = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 16) & 63) << 2, __float_as_int(x)));
    const float x2 = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 32) & 63) << 2, __float_as_int(x)));
    const float x3 = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 48) & 63) << 2, __float_as_int(x)));
    float s0 = init, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    asm volatile("s_nop 1" : : "v"(x), "v"(s1), "v"(s2), "v"(s3));   // 2 wait states between the VALU write of x and its first DPP read
    OT_BLOCK(x, KARR, 0);
    OT_BLOCK(x1, KARR, 1);
    OT_BLOCK(x2, KARR, 2);
    OT_BLOCK(x3, KARR, 3);
    return (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(64) void ot_kernel(RoitrOT a, unsigned long long* stats)
{
    __shared__ float T[64][65];
    __shared__ __attribute__((aligned(16))) float av[64];
    const int patch = blockIdx.x;
    const int lane = threadIdx.x;
    if (a.pair_off ? patch >= a.pair_off[a.pairs] : (patch % a.num_corr) >= a.n_corr[patch / a.num_corr]) return;
    float* out = a.out + (size_t)patch * OTN * OTN;
    const float alpha = *a.alpha;
    const float* sc = a.scores + (size_t)patch * 64 * 64;
    const bool rml = a.row_masks[(size_t)patch * 64 + lane] != 0, cml = a.col_masks[(size_t)patch * 64 + lane] != 0;
    const unsigned long long rbits = __ballot(rml), cbits = __ballot(cml);
    const int nvr = __popcll(rbits), nvc = __popcll(cbits);
    const float norm = -logf((float)nvr + (float)nvc);
    // column view (coalesced): lane = column l
#pragma unroll 8
    for (int i = 0; i < 64; ++i) T[i][lane] = sc[i * 64 + lane];
    __syncthreads();
    // row view: lane = row l.  m_l = row max over the 65 entries (masked entries are -1e6, the dustbin is alpha)
    float KR[64], KC[64];
    float m = rml ? alpha : -1e6f;
    float lo = alpha;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const bool ok = rml && ((cbits >> j) & 1);
        const float v = ok ? T[lane][j] : -1e6f;
        KR[j] = v;
        m = fmaxf(m, v);
        lo = ok ? fminf(lo, v) : lo;
    }
    // The exponential form keeps e^{v_j}, e^{u_i + m_i} and exp(S_ij - m_i) in fp32: it is exact-equivalent to the reference's
    // log-domain iteration only while those stay inside the fp32 range, i.e. while the scores of a row (dustbin score alpha
    // included) span a few tens -- what a network trained WITH this layer produces (alpha is learned against the scores).
    // Scores hundreds above alpha (every row wants the dustbin column to carry e^{200}) are left to ot_log_kernel: the corner
    // out[64][64] = NaN marks the patch (also set when the iteration ends non-finite after all).
    if (__ballot(rml && m - lo > OT_FAST_SPREAD) != 0) {
        if (lane == 0) out[64 * OTN + 64] = __int_as_float(0x7fc00000);
        return;
    }
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        KR[j] = (rml && ((cbits >> j) & 1)) ? expf(KR[j] - m) : 0.f;
        T[lane][j] = KR[j];
    }
    const float krd = rml ? expf(alpha - m) : 0.f;   // K'[l][64]
    const float kdr = cml ? 1.0f : 0.f;               // K'[64][l] = exp(alpha - m_64), m_64 = alpha
    const float kdd = 1.0f;                           // K'[64][64]
    __syncthreads();
    // rotated register order: slot 16 t + n of lane (R, c) pairs with vector element ((R + t) & 3) * 16 + ((c - n) & 15),
    // the element DPP row_ror:n delivers from row block t of the vector
    {
        const int R = lane >> 4, c = lane & 15;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int e = ((R + t) & 3) * 16 + ((c - n) & 15);
                KR[t * 16 + n] = T[lane][e];
                KC[t * 16 + n] = T[e][lane];
            }
    }
    // mu_i = exp(log_mu_i), nu likewise (modules.py:55-63)
    const float mu = rml ? expf(norm) : 0.f, nu = cml ? expf(norm) : 0.f;
    const float mu64 = expf(logf((float)nvc) + norm), nu64 = expf(logf((float)nvr) + norm);
    float al = 0.f, a64 = 0.f, bl = cml ? 1.0f : 0.f, b64 = 1.0f;   // v = 0
    for (int it = 0; it < a.num_iter; ++it) {
        // the dot products are cross-lane operations: evaluated by ALL lanes, outside the masked selects
        const float rs = ot_dot64(KR, bl, krd * b64, lane);
        al = rml ? mu / rs : 0.f;
        a64 = mu64 / (wave_sum(kdr * bl) + kdd * b64);
        const float cs = ot_dot64(KC, al, kdr * a64, lane);
        const float bn = cml ? nu / cs : 0.f;
        const float b64n = nu64 / (wave_sum(krd * al) + kdd * a64);
        const bool same = __ballot(__float_as_int(bn) != __float_as_int(bl)) == 0 && __float_as_int(b64n) == __float_as_int(b64);
        bl = bn; b64 = b64n;
        if (same) { if (stats && lane == 0) atomicAdd(stats + 1, (unsigned long long)(a.num_iter - 1 - it)); break; }
    }
    if (stats && lane == 0) atomicAdd(stats, 1ull);
    {   // a patch whose duals left the fp32 range after all goes to the log-domain kernel
        const bool bad = (rml && !(al > 0.f && al < INFINITY)) || (cml && !(bl > 0.f && bl < INFINITY)) || !(a64 > 0.f && a64 < INFINITY) ||
                         !(b64 > 0.f && b64 < INFINITY);
        if (__ballot(bad) != 0) {
            if (lane == 0) out[64 * OTN + 64] = __int_as_float(0x7fc00000);
            return;
        }
    }
    // outputs = S + u + v - norm  with u_i = log(a~_i) - m_i, v_j = log(b_j)  (modules.py:27,66-67)
    __syncthreads();
    const float ul = rml ? logf(al) - m : 0.f, vl = cml ? logf(bl) : 0.f;
    const float u64 = logf(a64) - alpha, v64 = logf(b64);
    av[lane] = ul;
    __syncthreads();
#pragma unroll 8
    for (int i = 0; i < 64; ++i) {
        const float sv = (cml && ((rbits >> i) & 1)) ? sc[i * 64 + lane] : -1e6f;
        out[i * OTN + lane] = sv + av[i] + vl - norm;
    }
    out[64 * OTN + lane] = (cml ? alpha : -1e6f) + u64 + vl - norm;
    out[lane * OTN + 64] = (rml ? alpha : -1e6f) + ul + v64 - norm;
    if (lane == 0) out[64 * OTN + 64] = alpha + u64 + v64 - norm;
}

// The reference's iteration verbatim, in the log domain (modules.py:21-27): u = log_mu - logsumexp_j(S + v), v = log_nu -
// logsumexp_i(S + u), masked entries and masked log_mu / log_nu at -1e6 like the reference, logsumexp = max + log(sum exp(. - max))
// in fp32.  Serves the patches ot_kernel declined (NaN in the corner of their output): any score range, ~20x the cost of the
// exponential form (130 exp per lane and iteration).  Same wave-per-patch layout and the same rotated register order, so the
// vector operand reaches the adds through DPP row rotations + three ds_bpermute row-block shifts.
#define OTL_ROT(X, N_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), 0x120 + (N_), 0xf, 0xf, false))
#define OTL_LOAD16(XA, SARR, VB, T_)                                                                                         \
    XA[(T_) * 16] = SARR[(T_) * 16] + VB;                                                                                     \
    XA[(T_) * 16 + 1] = SARR[(T_) * 16 + 1] + OTL_ROT(VB, 1);   XA[(T_) * 16 + 2] = SARR[(T_) * 16 + 2] + OTL_ROT(VB, 2);     \
    XA[(T_) * 16 + 3] = SARR[(T_) * 16 + 3] + OTL_ROT(VB, 3);   XA[(T_) * 16 + 4] = SARR[(T_) * 16 + 4] + OTL_ROT(VB, 4);     \
    XA[(T_) * 16 + 5] = SARR[(T_) * 16 + 5] + OTL_ROT(VB, 5);   XA[(T_) * 16 + 6] = SARR[(T_) * 16 + 6] + OTL_ROT(VB, 6);     \
    XA[(T_) * 16 + 7] = SARR[(T_) * 16 + 7] + OTL_ROT(VB, 7);   XA[(T_) * 16 + 8] = SARR[(T_) * 16 + 8] + OTL_ROT(VB, 8);     \
    XA[(T_) * 16 + 9] = SARR[(T_) * 16 + 9] + OTL_ROT(VB, 9);   XA[(T_) * 16 + 10] = SARR[(T_) * 16 + 10] + OTL_ROT(VB, 10); \
    XA[(T_) * 16 + 11] = SARR[(T_) * 16 + 11] + OTL_ROT(VB, 11); XA[(T_) * 16 + 12] = SARR[(T_) * 16 + 12] + OTL_ROT(VB, 12); \
    XA[(T_) * 16 + 13] = SARR[(T_) * 16 + 13] + OTL_ROT(VB, 13); XA[(T_) * 16 + 14] = SARR[(T_) * 16 + 14] + OTL_ROT(VB, 14); \
    XA[(T_) * 16 + 15] = SARR[(T_) * 16 + 15] + OTL_ROT(VB, 15)
// logsumexp over the 64 lane-distributed entries x (lane j holds element j) added to this lane's 64 scores SARR (rotated
// order), plus one extra term `extra`
__device__ __forceinline__ float ot_lse65(const float (&SARR)[64], float x, float extra, int lane)
{
    const float x1 = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 16) & 63) << 2, __float_as_int(x)));
    const float x2 = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 32) & 63) << 2, __float_as_int(x)));
    const float x3 = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 48) & 63) << 2, __float_as_int(x)));
    float X[64];
    OTL_LOAD16(X, SARR, x, 0);
    OTL_LOAD16(X, SARR, x1, 1);
    OTL_LOAD16(X, SARR, x2, 2);
    OTL_LOAD16(X, SARR, x3, 3);
    float mx = extra;
#pragma unroll
    for (int i = 0; i < 64; ++i) mx = fmaxf(mx, X[i]);
    float s0 = __expf(extra - mx), s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int i = 0; i < 64; i += 4) {
        s0 += __expf(X[i] - mx); s1 += __expf(X[i + 1] - mx); s2 += __expf(X[i + 2] - mx); s3 += __expf(X[i + 3] - mx);
    }
    return mx + __logf((s0 + s1) + (s2 + s3));
}
// logsumexp of one value per lane (64 lanes) and one extra term
__device__ __forceinline__ float ot_lse_wave(float x, float extra)
{
    const float mx = fmaxf(wave_max(x), extra);
    return mx + __logf(wave_sum(__expf(x - mx)) + __expf(extra - mx));
}

__global__ __launch_bounds__(64) void ot_log_kernel(RoitrOT a, unsigned long long* stats)
{
    __shared__ float T[64][65];
    const int patch = blockIdx.x;
    const int lane = threadIdx.x;
    if (a.pair_off ? patch >= a.pair_off[a.pairs] : (patch % a.num_corr) >= a.n_corr[patch / a.num_corr]) return;
    float* out = a.out + (size_t)patch * OTN * OTN;
    {
        const float corner = out[64 * OTN + 64];
        if (corner == corner) return;   // ot_kernel served this patch
    }
    if (stats && lane == 0) atomicAdd(stats + 2, 1ull);
    const float alpha = *a.alpha;
    const float NINF = -1e6f;
    const float* sc = a.scores + (size_t)patch * 64 * 64;
    const bool rml = a.row_masks[(size_t)patch * 64 + lane] != 0, cml = a.col_masks[(size_t)patch * 64 + lane] != 0;
    const unsigned long long rbits = __ballot(rml), cbits = __ballot(cml);
    const int nvr = __popcll(rbits), nvc = __popcll(cbits);
    const float norm = -logf((float)nvr + (float)nvc);
#pragma unroll 8
    for (int i = 0; i < 64; ++i) T[i][lane] = (cml && ((rbits >> i) & 1)) ? sc[i * 64 + lane] : NINF;   // padded_scores (modules.py:46-50)
    __syncthreads();
    float SR[64], SC[64];
    {
        const int R = lane >> 4, c = lane & 15;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int e = ((R + t) & 3) * 16 + ((c - n) & 15);
                SR[t * 16 + n] = T[lane][e];
                SC[t * 16 + n] = T[e][lane];
            }
    }
    const float srd = rml ? alpha : NINF;    // S[l][64]
    const float sdr = cml ? alpha : NINF;    // S[64][l]
    const float lmu = rml ? norm : NINF, lnu = cml ? norm : NINF;
    const float lmu64 = logf((float)nvc) + norm, lnu64 = logf((float)nvr) + norm;
    float ul = 0.f, u64 = 0.f, vl = 0.f, v64 = 0.f;
    for (int it = 0; it < a.num_iter; ++it) {
        ul = lmu - ot_lse65(SR, vl, srd + v64, lane);
        u64 = lmu64 - ot_lse_wave(sdr + vl, alpha + v64);
        const float vn = lnu - ot_lse65(SC, ul, sdr + u64, lane);
        v64 = lnu64 - ot_lse_wave(srd + ul, alpha + u64);
        vl = vn;
    }
    __syncthreads();
    T[0][lane] = ul;   // row 0 of the tile is free now: u by row index
    __syncthreads();
#pragma unroll 8
    for (int i = 0; i < 64; ++i) {
        const float sv = (cml && ((rbits >> i) & 1)) ? sc[i * 64 + lane] : NINF;
        out[i * OTN + lane] = sv + T[0][i] + vl - norm;
    }
    out[64 * OTN + lane] = sdr + u64 + vl - norm;
    out[lane * OTN + 64] = srd + ul + v64 - norm;
    if (lane == 0) out[64 * OTN + 64] = alpha + u64 + v64 - norm;
}

// ------------------------------------------------------------------ FineMatching (modules.py:216-324)
// one block per patch: mutual top-k on exp(score) (dustbin dropped, RIGA_v2.py:159-160), threshold, masks
// One wave owns a row (lane = column) and then a column (lane = row): the k best are peeled off with k wave maxima
// (value descending, lower index first among equals -- the rank order the element-wise count would give), the winners
// recorded as one 64-bit mask per row / column; the flag of (i, j) is then two bit tests.
__device__ __forceinline__ unsigned long long topk_mask(float v, int k, float conf)
{
    const int lane = threadIdx.x & 63;
    bool sel = false;
    for (int t = 0; t < k; ++t) {
        const float cand = sel ? -1.f : v;   // scores are exp(.) >= 0
        const float m = wave_max(cand);
        const unsigned long long eq = __ballot(cand == m);
        if (m >= 0.f && lane == __ffsll((long long)eq) - 1) sel = true;
    }
    return __ballot(sel && v > conf);
}

__global__ __launch_bounds__(256) void fine_flag_kernel(RoitrFine a)
{
    __shared__ float E[64][65];
    __shared__ unsigned long long rowm[64], colm[64];
    __shared__ int cnt_s[4];
    const int patch = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = a.limit;   // == 64 (checked by the launcher)
    unsigned char* fl = a.flags + (size_t)patch * L * L;
    uint4* fl16 = reinterpret_cast<uint4*>(fl) + tid;   // thread -> row tid/4, 16 columns from (tid%4)*16
    if (a.pair_off ? patch >= a.pair_off[a.pairs] : (patch % a.num_corr) >= a.n_corr[patch / a.num_corr]) {
        if (tid == 0) a.counts[patch] = 0;   // the emitter leaves on counts == 0 before it reads the flags of a dead patch
        return;
    }
    const float* sc = a.ot + (size_t)patch * (L + 1) * (L + 1);
    for (int e = tid; e < L * L; e += 256) { const int i = e >> 6, j = e & 63; E[i][j] = expf(sc[i * (L + 1) + j]); }
    __syncthreads();
    if (a.k <= 4) {
        // wave 0: lane = row, wave 1: lane = column.  One pass over the 64 entries keeps the k best in registers (strict >
        // against entries met earlier: the lower index stays ahead among equals, the order topk_mask peels them off in);
        // E[lane][j] / E[j][lane] are conflict-free (row pitch 65).  ~64 x 12 VALU per wave instead of 96 wave-wide maxima.
        if (wave < 2) {
            float bv[4] = {-1.f, -1.f, -1.f, -1.f};
            int bi[4] = {0, 0, 0, 0};
#pragma unroll 8
            for (int j = 0; j < 64; ++j) {
                float x = wave == 0 ? E[lane][j] : E[j][lane];
                int xi = j;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool up = x > bv[t];
                    const float tv = bv[t]; const int ti = bi[t];
                    bv[t] = up ? x : tv; bi[t] = up ? xi : ti;
                    x = up ? tv : x; xi = up ? ti : xi;
                }
            }
            unsigned long long m = 0ull;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < a.k && bv[t] >= 0.f && bv[t] > a.conf) m |= 1ull << bi[t];
            if (wave == 0) rowm[lane] = m; else colm[lane] = m;
        }
    } else {
        for (int r = wave; r < 64; r += 4) {
            const unsigned long long mr = topk_mask(E[r][lane], a.k, a.conf);
            const unsigned long long mc = topk_mask(E[lane][r], a.k, a.conf);
            if (lane == 0) { rowm[r] = mr; colm[r] = mc; }
        }
    }
    __syncthreads();
    const int i = tid >> 2, j0 = (tid & 3) * 16;
    const int* cm = a.col_masks + (size_t)patch * L;
    const bool row_ok = a.row_masks[(size_t)patch * L + i] != 0;
    const unsigned long long rmask = rowm[i];
    unsigned w[4] = {0, 0, 0, 0};
    int local = 0;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = j0 + jj;
        const bool rt = (rmask >> j) & 1, ct = (colm[j] >> i) & 1;
        const bool f = (a.mutual ? (rt && ct) : (rt || ct)) && row_ok && cm[j] != 0;
        w[jj >> 2] |= (f ? 1u : 0u) << (8 * (jj & 3));
        local += f ? 1 : 0;
    }
    *fl16 = make_uint4(w[0], w[1], w[2], w[3]);
    local = (int)wave_sum((float)local);
    if (lane == 0) cnt_s[wave] = local;
    __syncthreads();
    if (tid == 0) a.counts[patch] = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3];
}

// exclusive scan of per-patch counts (single block), total -> *n_out
// pair_starts (optional, pairs + 1 entries): first output row of every PAIR (+ the total): what a caller needs to split the
// correspondence list per pair without touching the per-patch offsets
__global__ __launch_bounds__(1024) void fine_scan_kernel(int n, const int* __restrict__ counts, int* __restrict__ offsets, int* __restrict__ n_out,
                                                         long out_cap, int pairs, int num_corr, const int* __restrict__ pair_off,
                                                         int* __restrict__ pair_starts)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int v = i < n ? counts[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wb = 0;
        for (int w = 0; w < wave; ++w) wb += wsum[w];
        const int carry = carry_s;
        // clamped like *n_out: with a caller-chosen out_cap below the true total the offsets of the patches past the cap all
        // read out_cap (their rows are not emitted), so offsets[i+1] - offsets[i] never exceeds what was written
        if (i < n) { const int o_ = carry + wb + incl - v; offsets[i] = (out_cap > 0 && o_ > out_cap) ? (int)out_cap : o_; }
        __syncthreads();
        if (tid == 1023) carry_s = carry + wb + incl;
        __syncthreads();
    }
    const int total = (out_cap > 0 && carry_s > out_cap) ? (int)out_cap : carry_s;
    if (tid == 0) *n_out = total;
    if (pair_starts) {
        __threadfence_block();
        for (int b = tid; b <= pairs; b += 1024) {
            const long first = pair_off ? (long)pair_off[b] : (long)b * num_corr;   // first patch slot of pair b (pairs: one past the last)
            pair_starts[b] = (b < pairs && first < n) ? offsets[first] : total;
        }
    }
}

// row-major (patch, i, j) compaction -- the order of torch.nonzero (modules.py:282)
__global__ __launch_bounds__(256) void fine_emit_kernel(RoitrFine a)
{
    __shared__ int wsum[4];
    const int patch = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = a.limit;
    if (a.counts[patch] == 0) return;
    const unsigned char* fl = a.flags + (size_t)patch * L * L;
    const float* sc = a.ot + (size_t)patch * (L + 1) * (L + 1);
    constexpr int per = 16;         // (L * L) / 256 consecutive entries per thread, L == 64 (checked by the launcher): ONE 16-byte load
    const uint4 f16 = reinterpret_cast<const uint4*>(fl)[tid];
    const unsigned fw[4] = {f16.x, f16.y, f16.z, f16.w};
    int c = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) c += __popc(fw[u]);   // a flag byte is 0 or 1
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int pos = a.offsets[patch] + incl - c;
    for (int w = 0; w < wave; ++w) pos += wsum[w];
    float g = 1.0f;
    if (a.global_scores) {   // coarse score of the patch's node pair: (pairs, num_corr), strided in both layouts
        const PatchSlot ps = patch_slot(patch, a.pairs, a.num_corr, a.n_corr, a.pair_off);
        g = a.global_scores[(size_t)ps.pair * a.num_corr + ps.p];
    }
    const long cap = a.out_cap > 0 ? a.out_cap : 0x7fffffffL;
#pragma unroll
    for (int u = 0; u < per; ++u) {
        const int e = tid * per + u;
        if (((fw[u >> 2] >> (8 * (u & 3))) & 1u) && pos < cap) {
            const int i = e / L, j = e % L;
            const float* rp = a.row_pts + ((size_t)patch * L + i) * 3;
            const float* cp = a.col_pts + ((size_t)patch * L + j) * 3;
            a.out_row_pts[(size_t)pos * 3] = rp[0]; a.out_row_pts[(size_t)pos * 3 + 1] = rp[1]; a.out_row_pts[(size_t)pos * 3 + 2] = rp[2];
            a.out_col_pts[(size_t)pos * 3] = cp[0]; a.out_col_pts[(size_t)pos * 3 + 1] = cp[1]; a.out_col_pts[(size_t)pos * 3 + 2] = cp[2];
            a.out_scores[pos] = expf(sc[i * (L + 1) + j]) * g;
            if (a.out_patch) a.out_patch[pos] = patch;
            ++pos;
        }
    }
}

}  // namespace

extern "C" int roitr_point_to_node_partition(int b, int n_points, int n_nodes, const float* pts, const int* pt_offset,
                                             const float* nodes, const int* node_offset, const int* cloud_of_node, int limit,
                                             int* p2n, float* p2n_dist, int* node_masks, int* knn_idx, int* knn_mask, hipStream_t stream)
{
    if (n_points <= 0 || n_nodes <= 0) return ROITR_OK;
    ROITR_HIP(hipMemsetAsync(node_masks, 0, sizeof(int) * (size_t)n_nodes, stream));
    p2n_assign_kernel<<<div_up(n_points, 256), 256, 0, stream>>>(n_points, pts, pt_offset, nullptr, nodes, node_offset, b, p2n, p2n_dist, node_masks);
    ROITR_LAUNCH_CHECK();
    p2n_topk_kernel<<<n_nodes, 256, 0, stream>>>(pt_offset, node_offset, cloud_of_node, p2n, p2n_dist, limit, knn_idx, knn_mask);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

// per pair: the (n_ref, n_src) score matrix, row / column sums, and the spill area of the chunked top-k (2 floats per
// 64-bit key; at most ceil(total / 16384) chunks x 1024 winners in the first round)
extern "C" size_t roitr_coarse_scratch_floats(int n_ref, int n_src)
{
    const size_t total = (size_t)n_ref * n_src;
    const size_t spill_keys = total > COARSE_LDS_KEYS ? ((total + COARSE_LDS_KEYS - 1) / COARSE_LDS_KEYS) * 1024 + 64 : 0;
    return total + n_ref + n_src + 16 + 2 * spill_keys;
}

extern "C" int roitr_coarse_matching(const RoitrCoarse* a, hipStream_t stream)
{
    if (a->pairs <= 0) return ROITR_OK;
    long cap = 1;
    while (cap < (long)a->max_ref * a->max_src) cap <<= 1;
    if (cap > COARSE_LDS_KEYS) cap = COARSE_LDS_KEYS;
    if (a->num_corr > 1024) return ROITR_ERR_UNSUPPORTED;   // winners per chunk of the chunked top-k
    ROITR_GRANT_LDS(coarse_match_kernel, 128 * 1024);
    RoitrCoarse c = *a;
    c.lds_cap = (int)cap;
    coarse_match_kernel<<<a->pairs, 1024, cap * 8, stream>>>(c);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_adaptive_matching(const RoitrCoarse* a, int min_num, float threshold, hipStream_t stream)
{
    if (a->pairs <= 0) return ROITR_OK;
    if (!a->xy) return ROITR_ERR_ARG;
    long cap = 1;
    while (cap < (long)a->max_ref * a->max_src) cap <<= 1;
    if (cap > COARSE_LDS_KEYS) cap = COARSE_LDS_KEYS;
    if (min_num > 1024) return ROITR_ERR_UNSUPPORTED;
    ROITR_GRANT_LDS(adaptive_match_kernel, 128 * 1024);
    RoitrCoarse c = *a;
    c.lds_cap = (int)cap;
    adaptive_match_kernel<<<a->pairs, 1024, cap * 8, stream>>>(c, min_num, threshold);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_patch_gather(const RoitrPatch* a, hipStream_t stream)
{
    const long total = (a->pair_off ? (long)a->slots : (long)a->pairs * a->num_corr) * a->limit;
    if (total <= 0) return ROITR_OK;
    patch_gather_kernel<<<div_up(total, 256), 256, 0, stream>>>(*a, total);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_patch_offsets(int pairs, const int* n_corr, int slots, int* pair_off, hipStream_t stream)
{
    if (pairs <= 0) return ROITR_OK;
    if (!n_corr || !pair_off || slots < 0) return ROITR_ERR_ARG;
    patch_offsets_kernel<<<1, 1024, 0, stream>>>(pairs, n_corr, slots, pair_off);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

namespace {
// one counter buffer, on the device that was current when counting was switched on; launches on other devices do not count
struct OtStats {
    unsigned long long* d = nullptr;
    int dev = -1;
    bool on = false;
    void enable(bool e)
    {
        if (e && !d) {
            if (hipGetDevice(&dev) != hipSuccess || hipMalloc(&d, 24) != hipSuccess) { d = nullptr; dev = -1; return; }
            (void)hipMemset(d, 0, 24);
        }
        on = e && d;
    }
    unsigned long long* for_current_device() const
    {
        int cur = -1;
        return on && hipGetDevice(&cur) == hipSuccess && cur == dev ? d : nullptr;
    }
};
OtStats& ot_stats() { static OtStats st; return st; }   // no HIP call at library load or unload
}  // namespace

/* Counters of the optimal-transport stage since the last reset (synchronous; diagnostics; not while a stream of the device is
 * capturing): enable = 1 switches counting on (on the current device) and zeroes the counters, 0 switches it off, -1 only reads.
 * out[0] live patches, out[1] Sinkhorn iterations skipped by the fixed-point exit, out[2] patches served by the log-domain kernel. */
extern "C" int roitr_ot_stats(int enable, unsigned long long* out)
{
    OtStats& S = ot_stats();
    if (out) {
        out[0] = out[1] = out[2] = 0;
        if (S.d) {
            ROITR_HIP(hipSetDevice(S.dev));
            ROITR_HIP(hipDeviceSynchronize());
            ROITR_HIP(hipMemcpy(out, S.d, 24, hipMemcpyDeviceToHost));
        }
    }
    if (enable >= 0) {
        S.enable(enable != 0);
        if (S.d && enable) { ROITR_HIP(hipDeviceSynchronize()); ROITR_HIP(hipMemset(S.d, 0, 24)); }
    }
    return ROITR_OK;
}

extern "C" int roitr_optimal_transport(const RoitrOT* a, hipStream_t stream)
{
    if (a->pairs <= 0) return ROITR_OK;
    if (a->limit != 64) return ROITR_ERR_UNSUPPORTED;
    const int patches = a->pair_off ? a->slots : a->pairs * a->num_corr;
    if (patches <= 0) return ROITR_OK;
    if (a->pair_off) roitr_prof_begin_live(ROITR_PROF_OT, (64.0 * 64 + 65.0 * 65) * 4.0, 0.0, a->pair_off + a->pairs, stream);   // live patches
    else roitr_prof_begin(ROITR_PROF_OT, (double)patches * (64.0 * 64 + 65.0 * 65) * 4.0, stream);
    // data-dependent work of this stage (roitr_ot_stats): live patches, Sinkhorn iterations skipped by the exact fixed-point exit,
    // patches the exponential form handed to the log-domain kernel
    unsigned long long* sd = ot_stats().for_current_device();
    ot_kernel<<<patches, 64, 0, stream>>>(*a, sd);
    ot_log_kernel<<<patches, 64, 0, stream>>>(*a, sd);   // the patches the exponential form declined; the others leave at once
    roitr_prof_end(ROITR_PROF_OT, stream);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_fine_matching(const RoitrFine* a, hipStream_t stream)
{
    if (a->pairs <= 0) return ROITR_OK;
    if (a->limit != 64) return ROITR_ERR_UNSUPPORTED;
    const int patches = a->pair_off ? a->slots : a->pairs * a->num_corr;
    if (patches <= 0) {
        ROITR_HIP(hipMemsetAsync(a->n_out, 0, sizeof(int), stream));
        if (a->pair_starts) ROITR_HIP(hipMemsetAsync(a->pair_starts, 0, sizeof(int) * ((size_t)a->pairs + 1), stream));
        return ROITR_OK;
    }
    fine_flag_kernel<<<patches, 256, 0, stream>>>(*a);
    ROITR_LAUNCH_CHECK();
    fine_scan_kernel<<<1, 1024, 0, stream>>>(patches, a->counts, a->offsets, a->n_out, a->out_cap, a->pairs, a->num_corr, a->pair_off, a->pair_starts);
    ROITR_LAUNCH_CHECK();
    fine_emit_kernel<<<patches, 256, 0, stream>>>(*a);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}
