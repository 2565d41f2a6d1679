// Exact k-nearest-neighbour query (+ fused point-pair features) for gfx950.
//
// Replaces knnquery_cuda_launcher (reference cpp_wrappers/pointops/src/knnquery/
// knnquery_cuda_kernel.cu:65-116) and, when asked, fuses pointops.queryandgroup's "k+1 then drop
// column 0" (functions/pointops.py:88-89) and lib/utils.py:358-389 calc_ppf_gpu into the same pass.
//
// Design (not a translation of the one-thread-per-query heap kernel):
//   * ONE WAVE PER QUERY.  M is a few thousand at most, so a thread-per-query launch leaves the
//     chip idle (5000 queries = 78 waves for 1024 SIMDs); a wave per query gives 5000 waves.
//   * The running best list is a SORTED list DISTRIBUTED ACROSS THE 64 LANES (position p lives in
//     lane p&63, register p>>6).  A batch of 64 candidates is evaluated with one coalesced load and
//     one fmaf chain per lane; `__ballot(d2 < tau)` prunes the batch; survivors are inserted one by
//     one with a scalar broadcast (v_readlane) + a one-lane shift (the list never leaves VGPRs).
//   * Output-sensitive search: a uniform grid (counting sort by cell, x-fastest so a row of cells
//     is one contiguous run) is built per cloud once per level; a query expands Chebyshev rings
//     until the (k+1)-th best distance is provably below everything unseen.  Small clouds are
//     scanned brute force in index order with the same list code.
//   * Exactness.  Distances use the oracle's arithmetic form (common.h sqdist3).  With pairwise
//     distinct distances among the best k+1 the reference's heap + heap_sort result is the unique
//     ascending list, which is what the sorted list holds.  When the best k+1 contain equal
//     distances the reference's answer depends on its heap history (reheap swaps on ==,
//     l.21-36; strict < admission, l.97): those queries (rare) are appended to a tie list and a
//     second kernel REPLAYS the reference algorithm for them exactly -- index-order scan, max-heap,
//     heap_sort -- wave-cooperatively (64 distances per step, ballot-pruned against the heap root).
#include "common.h"
#include "prof.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define KNN_FILL 1e10f  // knnquery_cuda_kernel.cu:89
#define GRID_MAX_CELLS 16384
#define GRID_MAX_DIM 255

struct RoitrGrid {  // one per cloud
    float ox, oy, oz, h, inv_h;
    int nx, ny, nz;
};

namespace {

__device__ __forceinline__ float rl_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ int rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// ---------------------------------------------------------------- lane-distributed sorted list
template <int NR>
struct WaveList {
    float d[NR];
    int i[NR];
    __device__ __forceinline__ void init(int fill_idx)
    {
#pragma unroll
        for (int r = 0; r < NR; ++r) { d[r] = KNN_FILL; i[r] = fill_idx; }
    }
    // value at list position p (wave-uniform p)
    __device__ __forceinline__ float dist_at(int p) const
    {
        float v = rl_f(d[0], p & 63);
        if (NR > 1) { const float v1 = rl_f(d[NR - 1], p & 63); v = p >= 64 ? v1 : v; }
        return v;
    }
    // insert wave-uniform (nd, ni) after every element <= nd; the last element falls off
    __device__ __forceinline__ void insert(float nd, int ni, int lane)
    {
        float carry_d = 0.f; int carry_i = 0; bool carry_gt = false;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const bool gt = d[r] > nd;
            float pd = __shfl_up(d[r], 1, 64);
            int pi = __shfl_up(i[r], 1, 64);
            bool pgt = pd > nd;
            if (lane == 0) { pd = carry_d; pi = carry_i; pgt = carry_gt; }
            if (NR > 1) {  // what register r+1's lane 0 will receive
                carry_d = rl_f(d[r], 63); carry_i = rl_i(i[r], 63); carry_gt = carry_d > nd;
            }
            d[r] = pgt ? pd : (gt ? nd : d[r]);
            i[r] = pgt ? pi : (gt ? ni : i[r]);
        }
    }
};

struct Query {
    float x, y, z;
};

// one batch of <= 64 candidates: lane holds (cd, ci); cd = +inf for idle lanes
template <int NR>
__device__ __forceinline__ void consume_batch(WaveList<NR>& L, float cd, int ci, float& tau, int tau_pos, int lane)
{
    unsigned long long m = __ballot(cd < tau);
    while (m) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        const float nd = rl_f(cd, l);
        if (nd < tau) {
            L.insert(nd, rl_i(ci, l), lane);
            tau = L.dist_at(tau_pos);
        }
    }
}

// PPF of (centre c with normal cn) vs (neighbour p with normal pn): common.h roitr_ppf4 (shared with ppf.hip)
#define ppf4 roitr_ppf4

struct KnnOut {
    int* idx;          // (m, nsample) or null
    float* dist2;      // (m, nsample) or null
    int* group_idx;    // (m, nsample-1): columns 1.. (queryandgroup), or null
    float* ppf;        // (m, nsample-1, 4) for columns 1.., or null
    const float* ref_normals;    // (n, 3), needed for ppf
    const float* query_normals;  // (m, 3), needed for ppf
    int* tie_count;
    int* tie_list;
    int b;             // number of clouds when the caller knows it (binary segment search), 0 = unknown (legacy entry point)
};

// writes one query's result row(s) from list-position-indexed (dist, idx) held in lanes
template <int NR>
__device__ __forceinline__ void write_rows(const KnnOut& o, int q, int nsample, const float (&d)[NR], const int (&i)[NR], int lane,
                                           const float* __restrict__ xyz, Query Q)
{
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int p = lane + 64 * r;
        if (p < nsample) {
            if (o.idx) o.idx[(size_t)q * nsample + p] = i[r];
            if (o.dist2) o.dist2[(size_t)q * nsample + p] = d[r];
            if (p >= 1) {
                if (o.group_idx) o.group_idx[(size_t)q * (nsample - 1) + p - 1] = i[r];
                if (o.ppf) {
                    const float* pp = xyz + (size_t)i[r] * 3;
                    const float* pn = o.ref_normals + (size_t)i[r] * 3;
                    const float* qn = o.query_normals + (size_t)q * 3;
                    const float4 f = ppf4(Q.x, Q.y, Q.z, qn[0], qn[1], qn[2], pp[0], pp[1], pp[2], pn[0], pn[1], pn[2]);
                    reinterpret_cast<float4*>(o.ppf)[(size_t)q * (nsample - 1) + p - 1] = f;
                }
            }
        }
    }
}

// equal neighbouring distances among positions [0, nsample] (i.e. the best nsample+1)?
template <int NR>
__device__ __forceinline__ bool has_tie(const WaveList<NR>& L, int nsample, int lane)
{
    bool t = false;
    float nxt_first = 0.f;
    if (NR > 1) nxt_first = rl_f(L.d[NR - 1], 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float nx = __shfl_down(L.d[r], 1, 64);
        const int p = lane + 64 * r;
        if (lane == 63) nx = (r + 1 < NR) ? nxt_first : __int_as_float(0x7fc00000);  // NaN: never equal
        t |= (p < nsample) && (L.d[r] == nx) && (L.d[r] < KNN_FILL);
    }
    return __ballot(t) != 0ull;
}

__device__ __forceinline__ void find_segment(int q, const int* __restrict__ offset, const int* __restrict__ new_offset, int& start,
                                             int& end, int& seg, int b = 0)
{
    int bt = 0;  // get_bt_idx, knnquery_cuda_kernel.cu:51-62 (a linear walk there; the batched engine has ~1000 clouds per call)
    if (b > 0) bt = segment_of(q, new_offset, b);
    else
    while (!(q < new_offset[bt])) bt++;
    start = bt == 0 ? 0 : offset[bt - 1];
    end = offset[bt];
    seg = bt;
}

template <int NR>
__device__ __forceinline__ void finish_query(WaveList<NR>& L, const KnnOut& o, int q, int nsample, int lane, const float* xyz, Query Q)
{
    if (has_tie(L, nsample, lane)) {
        if (lane == 0) {
            const int slot = atomicAdd(o.tie_count, 1);
            o.tie_list[slot] = q;
        }
        return;  // the replay kernel owns this query's rows
    }
    write_rows<NR>(o, q, nsample, L.d, L.i, lane, xyz, Q);
}

// ---------------------------------------------------------------- brute force, index order
template <int NR>
__global__ __launch_bounds__(256) void knn_brute_kernel(int m, int nsample, const float* __restrict__ xyz,
                                                        const float* __restrict__ new_xyz, const int* __restrict__ offset,
                                                        const int* __restrict__ new_offset, KnnOut o)
{
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= m) return;
    int start, end, seg;
    find_segment(q, offset, new_offset, start, end, seg, o.b);
    Query Q = {new_xyz[(size_t)q * 3], new_xyz[(size_t)q * 3 + 1], new_xyz[(size_t)q * 3 + 2]};
    WaveList<NR> L;
    L.init(start);
    const int tau_pos = nsample;  // keep the best nsample+1 exactly (tie detection)
    float tau = KNN_FILL;
    for (int base = start; base < end; base += 64) {
        const int k = base + lane;
        float cd = INFINITY;
        if (k < end) {
            const float* p = xyz + (size_t)k * 3;
            cd = sqdist3(Q.x, Q.y, Q.z, p[0], p[1], p[2]);
        }
        consume_batch<NR>(L, cd, k, tau, tau_pos, lane);
    }
    finish_query<NR>(L, o, q, nsample, lane, xyz, Q);
}

// ---------------------------------------------------------------- grid build (one workgroup per cloud)
// CAP: cells a cloud may get (LDS counters: 4 CAP bytes).  Small clouds are built with CAP = 4096 (16 KB instead of 64 KB of LDS:
// the kernel runs on the geometry stream beside the feature path, whose workgroups need the LDS -- round 4); the cell size
// adapts to the cap either way (the `tot <= CAP` loop below), cell_start keeps its GRID_MAX_CELLS + 1 stride per cloud.
template <int CAP>
__global__ __launch_bounds__(1024) void grid_build_kernel(const float* __restrict__ xyz, const int* __restrict__ offset,
                                                          RoitrGrid* __restrict__ grids, int* __restrict__ cell_start,
                                                          float4* __restrict__ sorted, float target_occupancy)
{
    __shared__ int cnt[CAP];
    __shared__ float red[6][16];
    __shared__ RoitrGrid G;
    __shared__ int wave_tot[16];
    const int c = blockIdx.x;
    const int start = c == 0 ? 0 : offset[c - 1];
    const int end = offset[c];
    const int n = end - start;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* cs = cell_start + (size_t)c * (GRID_MAX_CELLS + 1);

    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = start + tid; k < end; k += 1024) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[(size_t)k * 3 + a];
            mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = -wave_max(-mn[a]), hi = wave_max(mx[a]);
        if (lane == 0) { red[a][wave] = lo; red[3 + a][wave] = hi; }
    }
    __shared__ int occ_cells;
    __shared__ int refine;
    if (tid == 0) occ_cells = 0;
    __syncthreads();
    // cell size: from the bounding box first (h^3 = volume * target / n), then -- round 4 -- corrected by what the first histogram
    // says: a cloud is not its bounding box (a rotated cube fills a third of it, a scanned surface a few per cent), the cells that
    // hold points at all then carry 3 - 20 x the target and every query scans that many more candidates.  The second (third) pass
    // sizes the cells so that the OCCUPIED ones hold the target on average.
    auto choose = [&](float h_want) {   // thread 0: dims for a cell size (>= the 255-per-axis and GRID_MAX_CELLS limits)
        float lo[3], ext[3], me = 0.f;
        for (int a = 0; a < 3; ++a) {
            float l = red[a][0], h = red[3 + a][0];
            for (int w = 1; w < 16; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
            if (n == 0) { l = 0.f; h = 0.f; }
            lo[a] = l; ext[a] = h - l; me = fmaxf(me, ext[a]);
        }
        if (!(me > 0.f)) me = 1.f;
        float h = h_want;
        if (!(h > 0.f)) {
            float vol = 1.f;
            for (int a = 0; a < 3; ++a) vol *= fmaxf(ext[a], 1e-3f * me);
            h = cbrtf(vol * target_occupancy / (float)(n > 0 ? n : 1));
        }
        h = fmaxf(h, me / (float)GRID_MAX_DIM);
        int d[3];
        for (;;) {
            long tot = 1;
            for (int a = 0; a < 3; ++a) {
                int v = (int)floorf(ext[a] / h) + 1;
                d[a] = v < 1 ? 1 : (v > GRID_MAX_DIM ? GRID_MAX_DIM : v);
                tot *= d[a];
            }
            if (tot <= CAP) break;
            h *= 1.26f;
        }
        G.ox = lo[0]; G.oy = lo[1]; G.oz = lo[2]; G.h = h; G.inv_h = 1.0f / h; G.nx = d[0]; G.ny = d[1]; G.nz = d[2];
    };
    if (tid == 0) choose(0.f);
    __syncthreads();
    for (int k = tid; k < G.nx * G.ny * G.nz; k += 1024) cnt[k] = 0;   // only the cells in use (not all 16384 counters)
    __syncthreads();
    auto cell_of_g = [&](float x, float y, float z) {
        int cx = (int)floorf((x - G.ox) * G.inv_h), cy = (int)floorf((y - G.oy) * G.inv_h), cz = (int)floorf((z - G.oz) * G.inv_h);
        cx = min(max(cx, 0), G.nx - 1); cy = min(max(cy, 0), G.ny - 1); cz = min(max(cz, 0), G.nz - 1);
        return (cz * G.ny + cy) * G.nx + cx;
    };
    for (int pass = 0; pass < 3; ++pass) {
        for (int k = start + tid; k < end; k += 1024)
            atomicAdd(&cnt[cell_of_g(xyz[(size_t)k * 3], xyz[(size_t)k * 3 + 1], xyz[(size_t)k * 3 + 2])], 1);
        __syncthreads();
        if (pass == 2) break;
        const int nc_ = G.nx * G.ny * G.nz;
        int occ_l = 0;
        for (int k = tid; k < nc_; k += 1024) occ_l += cnt[k] > 0 ? 1 : 0;
        occ_l = (int)wave_sum((float)occ_l);       // <= 1024 * 16 / 16 per wave: exact in fp32
        if (lane == 0 && occ_l) atomicAdd(&occ_cells, occ_l);
        __syncthreads();
        if (tid == 0) {
            // points per occupied cell now vs the target: shrink the cells when they hold more than 1.3 x the target
            const float have = (float)n / (float)max(occ_cells, 1);
            refine = 0;
            if (n > 0 && have > 1.3f * target_occupancy) {
                const float h_old = G.h;
                choose(h_old * cbrtf(target_occupancy / have));
                refine = G.h < 0.97f * h_old ? 1 : 0;
                if (!refine) choose(h_old);
            }
            occ_cells = 0;
        }
        __syncthreads();
        if (!refine) break;
        for (int k = tid; k < G.nx * G.ny * G.nz; k += 1024) cnt[k] = 0;
        __syncthreads();
    }
    if (tid == 0) grids[c] = G;
    const RoitrGrid g = G;
    const int ncell = g.nx * g.ny * g.nz;
    auto cell_of = [&](float x, float y, float z) {
        int cx = (int)floorf((x - g.ox) * g.inv_h), cy = (int)floorf((y - g.oy) * g.inv_h), cz = (int)floorf((z - g.oz) * g.inv_h);
        cx = min(max(cx, 0), g.nx - 1); cy = min(max(cy, 0), g.ny - 1); cz = min(max(cz, 0), g.nz - 1);
        return (cz * g.ny + cy) * g.nx + cx;
    };
    // exclusive scan of cnt[0..ncell) : 16 cells per thread
    {
        constexpr int PER = CAP / 1024;
        int loc[PER], s = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { const int k = tid * PER + j; loc[j] = k < ncell ? cnt[k] : 0; s += loc[j]; }
        int incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
        int run = wbase + incl - s;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int k = tid * PER + j;
            if (k < ncell) { cnt[k] = run; cs[k] = start + run; }
            run += loc[j];
        }
        if (tid == 0) cs[ncell] = start + n;
    }
    __syncthreads();
    for (int k = start + tid; k < end; k += 1024) {
        const float x = xyz[(size_t)k * 3], y = xyz[(size_t)k * 3 + 1], z = xyz[(size_t)k * 3 + 2];
        const int pos = atomicAdd(&cnt[cell_of(x, y, z)], 1);
        sorted[start + pos] = make_float4(x, y, z, __int_as_float(k));
    }
}

// ---------------------------------------------------------------- grid query
// Rows (fixed cz, cy; cx in [x0, x1]) of the current shell are gathered into `runs` (one run per
// lane), prefix-summed across the wave, and the concatenated candidates are consumed 64 at a time.
template <int NR>
__global__ __launch_bounds__(256) void knn_grid_kernel(int m, int nsample, const float* __restrict__ xyz,
                                                       const float* __restrict__ new_xyz, const int* __restrict__ offset,
                                                       const int* __restrict__ new_offset, const RoitrGrid* __restrict__ grids,
                                                       const int* __restrict__ cell_start, const float4* __restrict__ sorted, KnnOut o)
{
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= m) return;
    int start, end, seg;
    find_segment(q, offset, new_offset, start, end, seg, o.b);
    const RoitrGrid g = grids[seg];
    const int* cs = cell_start + (size_t)seg * (GRID_MAX_CELLS + 1);
    Query Q = {new_xyz[(size_t)q * 3], new_xyz[(size_t)q * 3 + 1], new_xyz[(size_t)q * 3 + 2]};
    WaveList<NR> L;
    L.init(start);
    const int tau_pos = nsample;
    float tau = KNN_FILL;

    int c0[3];
    {
        const float t[3] = {(Q.x - g.ox) * g.inv_h, (Q.y - g.oy) * g.inv_h, (Q.z - g.oz) * g.inv_h};
        const int dim[3] = {g.nx, g.ny, g.nz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            // clamp in float first: far-away queries must not overflow the int conversion
            const float tc = fminf(fmaxf(t[a], 0.f), (float)(dim[a] - 1));
            c0[a] = (int)tc;
        }
    }
    const float margin = 2e-4f * g.h;
    const int maxr = max(max(g.nx, g.ny), g.nz);
    for (int r = 0; r <= maxr; ++r) {
        const int x0 = max(c0[0] - r, 0), x1 = min(c0[0] + r, g.nx - 1);
        const int y0 = max(c0[1] - r, 0), y1 = min(c0[1] + r, g.ny - 1);
        const int z0 = max(c0[2] - r, 0), z1 = min(c0[2] + r, g.nz - 1);
        const int ny = y1 - y0 + 1, nz = z1 - z0 + 1;
        // rows of this shell: (z,y) pairs; a row on the shell's z/y faces contributes [x0,x1],
        // an interior row contributes only the two x end cells (when they are new)
        const int nrows = ny * nz;
        for (int rb = 0; rb < nrows; rb += 32) {
            // lanes 0..31 -> row rb+lane, low piece; lanes 32..63 -> same row, high piece
            const int row = rb + (lane & 31);
            int rs = 0, re = 0;
            if (row < nrows) {
                const int cy = y0 + row % ny, cz = z0 + row / ny;
                const bool face = (r == 0) || (cy == c0[1] - r) || (cy == c0[1] + r) || (cz == c0[2] - r) || (cz == c0[2] + r);
                const int rowbase = (cz * g.ny + cy) * g.nx;
                if (face) {
                    if (lane < 32) { rs = cs[rowbase + x0]; re = cs[rowbase + x1 + 1]; }
                } else {
                    const int cx = lane < 32 ? c0[0] - r : c0[0] + r;
                    if (cx >= 0 && cx < g.nx) { rs = cs[rowbase + cx]; re = cs[rowbase + cx + 1]; }
                }
            }
            const int len = re - rs;
            int incl = len;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) { const int v = __shfl_up(incl, s, 64); if (lane >= s) incl += v; }
            const int total = rl_i(incl, 63);
            const int excl = incl - len;
            for (int b = 0; b < total; b += 64) {
                const int e = b + lane;
                // run holding element e = first lane whose inclusive prefix exceeds e (all lanes
                // run the same 6 probe steps so the cross-lane reads stay convergent)
                const int es = min(e, total - 1);
                int lo = 0, hi = 63;
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const int mid = (lo + hi) >> 1;
                    const bool gt = __shfl(incl, mid, 64) > es;
                    hi = gt ? mid : hi;
                    lo = gt ? lo : min(mid + 1, 63);
                }
                const int src_run = __shfl(rs, lo, 64) + (es - __shfl(excl, lo, 64));
                const int src = e < total ? src_run : -1;
                float cd = INFINITY; int ci = 0;
                if (src >= 0) {
                    const float4 p = sorted[src];
                    cd = sqdist3(Q.x, Q.y, Q.z, p.x, p.y, p.z);
                    ci = __float_as_int(p.w);
                }
                consume_batch<NR>(L, cd, ci, tau, tau_pos, lane);
            }
        }
        // everything unseen lies outside the searched box: lower-bound its distance
        float dmin = INFINITY;
        if (x0 > 0) dmin = fminf(dmin, Q.x - __fmaf_rn((float)x0, g.h, g.ox));
        if (x1 < g.nx - 1) dmin = fminf(dmin, __fmaf_rn((float)(x1 + 1), g.h, g.ox) - Q.x);
        if (y0 > 0) dmin = fminf(dmin, Q.y - __fmaf_rn((float)y0, g.h, g.oy));
        if (y1 < g.ny - 1) dmin = fminf(dmin, __fmaf_rn((float)(y1 + 1), g.h, g.oy) - Q.y);
        if (z0 > 0) dmin = fminf(dmin, Q.z - __fmaf_rn((float)z0, g.h, g.oz));
        if (z1 < g.nz - 1) dmin = fminf(dmin, __fmaf_rn((float)(z1 + 1), g.h, g.oz) - Q.z);
        if (dmin == INFINITY) break;  // whole grid covered
        const float dm = dmin - margin;
        if (dm > 0.f && tau < dm * dm) break;
    }
    finish_query<NR>(L, o, q, nsample, lane, xyz, Q);
}

// ---------------------------------------------------------------- grid query with buffered selection (34 <= nsample + 1 <= 128)
// For large k the one-at-a-time insertion of knn_grid_kernel (and the 66/101-slot register chains of knn_lane_kernel)
// cost ~30 instructions per accepted candidate and almost every candidate of the first shells is accepted.  Here the
// candidates of a shell that beat the current (nsample+1)-th distance are appended to an LDS buffer by ballot
// compaction; once the buffer holds nsample+1 entries it is cut down to the <= 128 smallest by ONE histogram pass
// (64 distance bins in LDS, prefix scan across the lanes, everything up to the bin that contains rank nsample+1 is
// kept), and those are sorted by a 128-element bitonic network held two per lane.  Exactness: the selection keeps a
// superset of the best nsample+1, the sort orders them by (distance, index), ties among the best nsample+1 go to the
// replay kernel exactly as in the other kernels; a query whose buffers would overflow goes there as well.
constexpr int SEL_CAP = 1024;

// value of lane ^ J: DPP inside the 16-lane rows (quad permutes, row rotates: a few cycles, no LDS), ds_bpermute across
template <int J>
__device__ __forceinline__ int lane_xor(int v, int lane)
{
    if (J == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);        // quad_perm [1,0,3,2]
    if (J == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);        // quad_perm [2,3,0,1]
    if (J == 4) {   // two bank-masked row rotates into one register: banks 1, 3 (lane & 4) read lane - 4, banks 0, 2 read lane + 4
        const int dn = __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xa, false);      // row_ror:4  -> from lane - 4
        return __builtin_amdgcn_update_dpp(dn, v, 0x12C, 0xf, 0x5, false);             // row_ror:12 -> from lane + 4
    }
    if (J == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);       // row_ror:8  -> lane ^ 8
    return __shfl_xor(v, J, 64);
}

// sort the 128 (distance, index) pairs of a WaveList<2> ascending; element e = lane + 64 r
template <int K, int J>
__device__ __forceinline__ void sort128_step(WaveList<2>& L, int lane)
{
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int e = lane + 64 * r;
        const float pd = __int_as_float(lane_xor<J>(__float_as_int(L.d[r]), lane));
        const int pi = lane_xor<J>(L.i[r], lane);
        const bool up = (e & K) == 0;          // ascending block
        const bool lower = (lane & J) == 0;    // this lane holds the lower position of the pair
        // distances are >= 0: their bit patterns order like unsigned integers, so (distance, index) is one 64-bit compare
        const unsigned long long mk_ = ((unsigned long long)__float_as_uint(L.d[r]) << 32) | (unsigned)L.i[r];
        const unsigned long long pk_ = ((unsigned long long)__float_as_uint(pd) << 32) | (unsigned)pi;
        const bool mine_gt = mk_ > pk_;
        const bool take = (lower == up) ? mine_gt : !mine_gt;   // min to the lower position of ascending blocks
        L.d[r] = take ? pd : L.d[r];
        L.i[r] = take ? pi : L.i[r];
    }
}
template <int K>
__device__ __forceinline__ void sort128_stage(WaveList<2>& L, int lane)
{
    if (K >= 128) {   // partner = the lane's other register, ascending
        const bool sw = (L.d[0] > L.d[1]) || (L.d[0] == L.d[1] && L.i[0] > L.i[1]);
        const float td = sw ? L.d[1] : L.d[0]; const int ti = sw ? L.i[1] : L.i[0];
        L.d[1] = sw ? L.d[0] : L.d[1]; L.i[1] = sw ? L.i[0] : L.i[1];
        L.d[0] = td; L.i[0] = ti;
    }
    if (K >= 64) sort128_step<K, 32>(L, lane);
    if (K >= 32) sort128_step<K, 16>(L, lane);
    if (K >= 16) sort128_step<K, 8>(L, lane);
    if (K >= 8) sort128_step<K, 4>(L, lane);
    if (K >= 4) sort128_step<K, 2>(L, lane);
    sort128_step<K, 1>(L, lane);
}
__device__ __forceinline__ void wave_sort128(WaveList<2>& L, int lane)
{
    sort128_stage<2>(L, lane); sort128_stage<4>(L, lane); sort128_stage<8>(L, lane); sort128_stage<16>(L, lane);
    sort128_stage<32>(L, lane); sort128_stage<64>(L, lane); sort128_stage<128>(L, lane);
}
// qlist != nullptr: the queries are qlist[0 .. *qcount) (the ones knn_cell_kernel handed over), walked with a grid stride.
__global__ __launch_bounds__(256) void knn_gridsel_kernel(int m, int nsample, const float* __restrict__ xyz,
                                                          const float* __restrict__ new_xyz, const int* __restrict__ offset,
                                                          const int* __restrict__ new_offset, const RoitrGrid* __restrict__ grids,
                                                          const int* __restrict__ cell_start, const float4* __restrict__ sorted, KnnOut o,
                                                          const int* __restrict__ qlist, const int* __restrict__ qcount)
{
    __shared__ float bd_[4][SEL_CAP];
    __shared__ int bi_[4][SEL_CAP];
    __shared__ float ld_[4][128];
    __shared__ int li_[4][128];
    __shared__ int hist_[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n_q = qlist ? *qcount : m;
    for (int t_ = blockIdx.x * 4 + w; t_ < n_q; t_ += gridDim.x * 4) {
    const int q = qlist ? qlist[t_] : t_;
    float* bd = bd_[w]; int* bi = bi_[w]; float* ld = ld_[w]; int* li = li_[w]; int* hist = hist_[w];
    int start, end, seg;
    find_segment(q, offset, new_offset, start, end, seg, o.b);
    const RoitrGrid g = grids[seg];
    const int* cs = cell_start + (size_t)seg * (GRID_MAX_CELLS + 1);
    Query Q = {new_xyz[(size_t)q * 3], new_xyz[(size_t)q * 3 + 1], new_xyz[(size_t)q * 3 + 2]};
    WaveList<2> L;
    L.init(start);
    const int need = nsample + 1;   // keep the best nsample+1 exactly (tie detection)
    float tau = KNN_FILL;
    int cnt = 0;                    // entries in the LDS buffer (wave-uniform)
    bool overflow = false, sorted_ok = false;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    int c0[3];
    {
        const float t[3] = {(Q.x - g.ox) * g.inv_h, (Q.y - g.oy) * g.inv_h, (Q.z - g.oz) * g.inv_h};
        const int dim[3] = {g.nx, g.ny, g.nz};
#pragma unroll
        for (int a = 0; a < 3; ++a) c0[a] = (int)fminf(fmaxf(t[a], 0.f), (float)(dim[a] - 1));
    }
    // buffer -> the <= 128 best in L, sorted; buffer front rewritten with them
    auto reduce = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int n_in = cnt;
        const float* sd = bd; const int* si = bi;
        if (cnt > 128) {
            float mn = INFINITY, mx = -INFINITY;
            for (int e = lane; e < cnt; e += 64) { const float v = bd[e]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
            mn = -wave_max(-mn); mx = wave_max(mx);
            if (!(mx > mn)) { overflow = true; return; }
            const float scale = 63.999f / (mx - mn);
            hist[lane] = 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int e = lane; e < cnt; e += 64) atomicAdd(&hist[min(63, (int)((bd[e] - mn) * scale))], 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            int incl = hist[lane];
#pragma unroll
            for (int s_ = 1; s_ < 64; s_ <<= 1) { const int v = __shfl_up(incl, s_, 64); if (lane >= s_) incl += v; }
            const unsigned long long ge = __ballot(incl >= need);
            const int B = __ffsll((long long)ge) - 1;     // ge != 0: cnt >= need here
            const int nsel = rl_i(incl, B);
            if (nsel > 128) { overflow = true; return; }
            int base = 0;
            for (int e0 = 0; e0 < cnt; e0 += 64) {
                const int e = e0 + lane;
                const float v = e < cnt ? bd[e] : 0.f;
                const bool keep = e < cnt && min(63, (int)((v - mn) * scale)) <= B;
                const unsigned long long mk = __ballot(keep);
                if (keep) { const int pos = base + __popcll(mk & lt_mask); ld[pos] = v; li[pos] = bi[e]; }
                base += __popcll(mk);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            n_in = nsel; sd = ld; si = li;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int e = lane + 64 * r;
            L.d[r] = e < n_in ? sd[e] : KNN_FILL;
            L.i[r] = e < n_in ? si[e] : start;
        }
        wave_sort128(L, lane);
        // keep only what can still matter: the best `need` (everything behind is >= tau)
        cnt = min(n_in, need);
#pragma unroll
        for (int r = 0; r < 2; ++r) { const int e = lane + 64 * r; if (e < cnt) { bd[e] = L.d[r]; bi[e] = L.i[r]; } }
        tau = cnt >= need ? L.dist_at(nsample) : KNN_FILL;
        sorted_ok = true;
    };

    const float margin = 2e-4f * g.h;
    const int maxr = max(max(g.nx, g.ny), g.nz);
    for (int r = 0; r <= maxr && !overflow; ++r) {
        const int x0 = max(c0[0] - r, 0), x1 = min(c0[0] + r, g.nx - 1);
        const int y0 = max(c0[1] - r, 0), y1 = min(c0[1] + r, g.ny - 1);
        const int z0 = max(c0[2] - r, 0), z1 = min(c0[2] + r, g.nz - 1);
        // rows of this shell, wave-uniform: a row on a y/z face contributes the run of cells [x0, x1], an interior row only
        // its two (new) x end cells.  Every lane strides a run -- no cross-lane traffic in the enumeration.
        // (Measured and dropped: issuing the first loads of 8 runs together, 7.9 vs 6.7 ms on the k = 64 stress case --
        // the kernel is VALU-bound (SQ: 47 % active, 2800 VALU + 1600 SALU instructions per query), not load-latency-bound.)
        auto scan_run = [&](int rs, int re) {
            for (int p0 = rs; p0 < re && !overflow; p0 += 64) {
                const int p = p0 + lane;
                bool pass = false; float cd = 0.f; int ci = 0;
                if (p < re) {
                    const float4 c = sorted[p];
                    cd = sqdist3(Q.x, Q.y, Q.z, c.x, c.y, c.z);
                    ci = __float_as_int(c.w);
                    pass = cd < tau;
                }
                const unsigned long long mk = __ballot(pass);
                const int add = __popcll(mk);
                if (cnt + add > SEL_CAP) { overflow = true; break; }
                if (pass) { const int pos = cnt + __popcll(mk & lt_mask); bd[pos] = cd; bi[pos] = ci; }
                cnt += add;
                sorted_ok = sorted_ok && add == 0;
            }
        };
        for (int cz = z0; cz <= z1 && !overflow; ++cz) {
            for (int cy = y0; cy <= y1 && !overflow; ++cy) {
                const int rowbase = (cz * g.ny + cy) * g.nx;
                const bool face = (r == 0) || (cy == c0[1] - r) || (cy == c0[1] + r) || (cz == c0[2] - r) || (cz == c0[2] + r);
                if (face) scan_run(cs[rowbase + x0], cs[rowbase + x1 + 1]);
                else {
                    const int xa = c0[0] - r, xb = c0[0] + r;
                    if (xa >= 0) scan_run(cs[rowbase + xa], cs[rowbase + xa + 1]);
                    if (xb < g.nx) scan_run(cs[rowbase + xb], cs[rowbase + xb + 1]);
                }
            }
        }
        if (overflow) break;
        if (cnt >= need && !sorted_ok) reduce();
        if (overflow) break;
        float dmin = INFINITY;
        if (x0 > 0) dmin = fminf(dmin, Q.x - __fmaf_rn((float)x0, g.h, g.ox));
        if (x1 < g.nx - 1) dmin = fminf(dmin, __fmaf_rn((float)(x1 + 1), g.h, g.ox) - Q.x);
        if (y0 > 0) dmin = fminf(dmin, Q.y - __fmaf_rn((float)y0, g.h, g.oy));
        if (y1 < g.ny - 1) dmin = fminf(dmin, __fmaf_rn((float)(y1 + 1), g.h, g.oy) - Q.y);
        if (z0 > 0) dmin = fminf(dmin, Q.z - __fmaf_rn((float)z0, g.h, g.oz));
        if (z1 < g.nz - 1) dmin = fminf(dmin, __fmaf_rn((float)(z1 + 1), g.h, g.oz) - Q.z);
        if (dmin == INFINITY) break;  // whole grid covered
        const float dm = dmin - margin;
        if (dm > 0.f && tau < dm * dm) break;
    }
    if (!overflow && !sorted_ok) reduce();   // fewer than nsample+1 candidates in the whole cloud, or a trailing append
    if (overflow) {   // rare: hand the query to the exact replay
        if (lane == 0) { const int slot = atomicAdd(o.tie_count, 1); o.tie_list[slot] = q; }
        continue;
    }
    finish_query<2>(L, o, q, nsample, lane, xyz, Q);
    }
}

// ---------------------------------------------------------------- cell kernel: large k, self queries (BASELINE config 5)
// One WORKGROUP PER GRID CELL.  The 3 x 3 x 3 cell neighbourhood of the cell (9 contiguous runs of the counting-sorted
// array) is staged ONCE into LDS and shared by all queries of the cell; a wave takes one query at a time:
//   A. distances to every staged candidate (conflict-free ds_read_b128, 64 per step); only candidates inside the
//      GUARANTEE radius (everything unseen is farther than the distance dm to the faces of the 3 x 3 x 3 box) can belong
//      to the answer -- those (~20 %) are ballot-compacted into a per-wave LDS list;
//   B. exact selection of the S = nsample - 1 nearest OTHER points (the query itself is position 0 of its row): a 64-bin
//      histogram over [0, dm^2) finds the bin that holds rank S, the few elements of that bin are ranked against each
//      other, which gives the exact threshold tau and tells whether the cut falls between two EQUAL distances;
//   C. the S elements <= tau are compacted one per lane and ordered by a 64-lane bitonic network (21 steps, DPP inside
//      the 16-lane rows), neighbour coordinates for the fused PPF come from the staged LDS copy.
// No per-candidate list insertion anywhere (the old kernels spend ~2.8 k VALU instructions per query on it at k = 64).
// Exactness: same fp32 distance form (sqdist3); a tie anywhere among the best nsample + 1 (equal neighbours after the
// sort, a second point at distance 0, a tie at the cut) sends the query to the replay kernel like every other kernel;
// anything this kernel cannot decide from its staged data (guarantee radius not reached, neighbourhood larger than the
// LDS budget, cloud smaller than nsample) is handed to knn_gridsel_kernel through a retry list.
constexpr int CELL_CAP = 1024;    // staged candidates per cell (16 KB)
constexpr int CELL_NEAR = 256;    // candidates inside the guarantee radius kept per query (4 per lane)

// lanes whose pair keeps the SMALLER key: the lower position of an ascending block, the upper position of a descending one
constexpr unsigned long long sort64_mask(int K, int J)
{
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) m |= (unsigned long long)((((l & J) == 0) == ((l & K) == 0)) ? 1 : 0) << l;
    return m;
}
// One compare per step (round 6: 5 VALU + 2 SALU instead of 13 + spilled lane masks): `d > pd` as a lane mask, flipped by a literal
// for the lanes that keep the larger key.  Two EQUAL keys: the lane that keeps the larger one takes its partner's pair, so the pair
// (d, j_a) exists twice and (d, j_b) is gone -- the caller's equal-neighbours check after the sort sees the two d side by side and
// hands the query to the replay, which is where a query with equal distances among its best goes anyway.
template <int K, int J>
__device__ __forceinline__ void sort64_step(float& d, int& j, int lane)
{
    constexpr unsigned long long M = sort64_mask(K, J);
    const float pd = __int_as_float(lane_xor<J>(__float_as_int(d), lane));
    const int pj = lane_xor<J>(j, lane);
    const bool take = __builtin_amdgcn_inverse_ballot_w64(~(__ballot(d > pd) ^ M));
    d = take ? pd : d;
    j = take ? pj : j;
}
template <int K>
__device__ __forceinline__ void sort64_stage(float& d, int& j, int lane)
{
    if (K >= 64) sort64_step<K, 32>(d, j, lane);
    if (K >= 32) sort64_step<K, 16>(d, j, lane);
    if (K >= 16) sort64_step<K, 8>(d, j, lane);
    if (K >= 8) sort64_step<K, 4>(d, j, lane);
    if (K >= 4) sort64_step<K, 2>(d, j, lane);
    sort64_step<K, 1>(d, j, lane);
}
__device__ __forceinline__ void wave_sort64(float& d, int& j, int lane)
{
    sort64_stage<2>(d, j, lane); sort64_stage<4>(d, j, lane); sort64_stage<8>(d, j, lane);
    sort64_stage<16>(d, j, lane); sort64_stage<32>(d, j, lane); sort64_stage<64>(d, j, lane);
}
__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
    // Kogge-Stone inside the 16-lane rows (row_shr, zero fill), then the row totals: row_bcast:15 into rows 1 and 3, row_bcast:31 into 2 - 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

__global__ __launch_bounds__(256) void knn_cell_kernel(int nsample, const float* __restrict__ xyz, const int* __restrict__ offset,
                                                       const RoitrGrid* __restrict__ grids, const int* __restrict__ cell_start,
                                                       const float4* __restrict__ sorted, KnnOut o, int* __restrict__ retry_count,
                                                       int* __restrict__ retry_list)
{
    __shared__ float4 cand[CELL_CAP];
    __shared__ uint2 nl_[4][CELL_NEAR];      // per wave: (distance bits, staged slot) of the candidates inside the limit
    __shared__ int hist_[4][64];
    __shared__ float sd_[4][64];
    __shared__ int sj_[4][64];
    __shared__ int run_s[9], run_off[10];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: everything derived from it stays scalar
    const int seg = blockIdx.y;
    const RoitrGrid g = grids[seg];
    const int* cs = cell_start + (size_t)seg * (GRID_MAX_CELLS + 1);
    const int ncell = g.nx * g.ny * g.nz;
    const int S = nsample - 1;                 // neighbours other than the query itself (<= 64)
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint2* nl = nl_[w]; int* hist = hist_[w]; float* sd = sd_[w]; int* sj = sj_[w];
    const float margin = 2e-4f * g.h;
    auto retry = [&](int q) { if (lane == 0) { const int slot = atomicAdd(retry_count, 1); retry_list[slot] = q; } };
    auto tie = [&](int q) { if (lane == 0) { const int slot = atomicAdd(o.tie_count, 1); o.tie_list[slot] = q; } };

    for (int cell = blockIdx.x; cell < ncell; cell += gridDim.x) {
        const int qs = cs[cell], qe = cs[cell + 1];
        if (qe == qs) continue;                 // block-uniform
        const int cx = cell % g.nx, cy = (cell / g.nx) % g.ny, cz = cell / (g.nx * g.ny);
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
        const int y0 = max(cy - 1, 0), y1 = min(cy + 1, g.ny - 1);
        const int z0 = max(cz - 1, 0), z1 = min(cz + 1, g.nz - 1);
        __syncthreads();                        // the previous cell's queries are done with cand / run tables
        if (tid < 64) {   // lanes 0..8: one (z, y) row of the neighbourhood each; exclusive prefix of the run lengths across them
            int a = 0, b = 0;
            if (tid < 9) {
                const int yy = cy - 1 + tid % 3, zz = cz - 1 + tid / 3;
                if (yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
                    const int rowbase = (zz * g.ny + yy) * g.nx;
                    a = cs[rowbase + x0]; b = cs[rowbase + x1 + 1];
                }
            }
            const int incl = wave_incl_scan(b - a, lane);
            if (tid < 9) { run_s[tid] = a; run_off[tid] = incl - (b - a); }
            if (tid == 8) run_off[9] = incl;
        }
        __syncthreads();
        const int C = run_off[9];
        if (C > CELL_CAP) {                     // neighbourhood does not fit: all queries of the cell take the general path
            for (int p = qs + tid; p < qe; p += 256) { const int slot = atomicAdd(retry_count, 1); retry_list[slot] = __float_as_int(sorted[p].w); }
            continue;
        }
        for (int e = tid; e < ((C + 63) & ~63); e += 256) {
            int r = 0;
#pragma unroll
            for (int u = 1; u < 9; ++u) r += (e >= run_off[u]) ? 1 : 0;
            // the tail of the last 64-group: points at infinity, whose distance passes no limit (no validity mask in the scan)
            cand[e] = e < C ? sorted[run_s[r] + (e - run_off[r])] : make_float4(INFINITY, INFINITY, INFINITY, 0.f);
        }
        __syncthreads();
        const int centre0 = run_off[4] - run_s[4];   // slot of sorted[p] (p in the centre row run) = p + centre0
        const int T = (C + 63) >> 6;
        // the density radius of the first pass depends on the cell only (round 6: it was a powf per query, ~6 % of the kernel's VALU work)
        const float rho_loc = (float)C / (float)((x1 - x0 + 1) * (y1 - y0 + 1) * (z1 - z0 + 1));
        const float rcap2 = g.h * g.h * powf(2.5f * (float)S / (4.18879f * fmaxf(rho_loc, 1e-3f)), 0.6667f);

        for (int p = qs + w; p < qe; p += 4) {
            const int qslot = p + centre0;
            const float4 Qp = cand[qslot];
            const int q = __float_as_int(Qp.w);
            // guarantee radius: distance to the nearest face of the searched box that has unseen cells behind it
            float dmin = INFINITY;
            if (x0 > 0) dmin = fminf(dmin, Qp.x - __fmaf_rn((float)x0, g.h, g.ox));
            if (x1 < g.nx - 1) dmin = fminf(dmin, __fmaf_rn((float)(x1 + 1), g.h, g.ox) - Qp.x);
            if (y0 > 0) dmin = fminf(dmin, Qp.y - __fmaf_rn((float)y0, g.h, g.oy));
            if (y1 < g.ny - 1) dmin = fminf(dmin, __fmaf_rn((float)(y1 + 1), g.h, g.oy) - Qp.y);
            if (z0 > 0) dmin = fminf(dmin, Qp.z - __fmaf_rn((float)z0, g.h, g.oz));
            if (z1 < g.nz - 1) dmin = fminf(dmin, __fmaf_rn((float)(z1 + 1), g.h, g.oz) - Qp.z);
            const bool covered = dmin == INFINITY;          // the box is the whole grid
            const float dm = dmin - margin;
            if (!covered && !(dm > 0.f)) { retry(q); continue; }
            // the list has CELL_NEAR slots: the first pass looks no farther than the radius that holds ~2.5 S points at the local
            // density (a query in the middle of its cell has dm = 1.5 h: 370 points at 26 per cell), the guarantee radius at most
            float lim = covered ? rcap2 : fminf(dm * dm, rcap2);

            // ---- A: distances, compaction of the candidates inside the guarantee radius.  Four 64-candidate groups per round: the
            // four LDS reads are in flight together (round 6; one read, one wait per group before); the list position is one v_mbcnt
            // pair on the ballot.  The query ITSELF (distance 0) is listed like any candidate -- cnt counts it, the rank search below
            // looks for rank S + 1, the selection drops it by its slot -- so the scan carries no per-group mask at all.
            int cnt = 0;
            auto scan_staged = [&]() {
                cnt = 0;
#pragma unroll 1
                for (int t0 = 0; t0 < T; t0 += 4) {
                    float4 c[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) c[u] = cand[min(t0 + u, CELL_CAP / 64 - 1) * 64 + lane];   // past C: never counted
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int t = t0 + u;
                        if (t < T) {
                            const float d = sqdist3(Qp.x, Qp.y, Qp.z, c[u].x, c[u].y, c[u].z);
                            const bool near = d < lim;
                            const unsigned long long mk = __ballot(near);
                            const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, cnt));
                            if (near && pos < CELL_NEAR) nl[pos] = make_uint2(__float_as_uint(d), (unsigned)(t * 64 + lane));
                            cnt += __popcll(mk);
                        }
                    }
                }
            };
            scan_staged();
            if (cnt > CELL_NEAR) { retry(q); continue; }   // too dense for the list
            if (cnt <= S) {   // cnt counts the query itself
                // Guarantee radius not reached (5.5 % of the queries of a uniform cube: the cells at its faces, whose sphere is cut
                // by the boundary).  Second ring for THIS query: the 5 x 5 x 5 box's guarantee radius, capped at the radius the
                // local density says holds ~1.7 S neighbours (the list has CELL_NEAR slots); the staged 27 cells are re-scanned
                // from LDS with the new limit, the 98 shell cells come straight from the sorted array (slot = -(position + 1)).
                const int x0b = max(cx - 2, 0), x1b = min(cx + 2, g.nx - 1);
                const int y0b = max(cy - 2, 0), y1b = min(cy + 2, g.ny - 1);
                const int z0b = max(cz - 2, 0), z1b = min(cz + 2, g.nz - 1);
                float dmin2 = INFINITY;
                if (x0b > 0) dmin2 = fminf(dmin2, Qp.x - __fmaf_rn((float)x0b, g.h, g.ox));
                if (x1b < g.nx - 1) dmin2 = fminf(dmin2, __fmaf_rn((float)(x1b + 1), g.h, g.ox) - Qp.x);
                if (y0b > 0) dmin2 = fminf(dmin2, Qp.y - __fmaf_rn((float)y0b, g.h, g.oy));
                if (y1b < g.ny - 1) dmin2 = fminf(dmin2, __fmaf_rn((float)(y1b + 1), g.h, g.oy) - Qp.y);
                if (z0b > 0) dmin2 = fminf(dmin2, Qp.z - __fmaf_rn((float)z0b, g.h, g.oz));
                if (z1b < g.nz - 1) dmin2 = fminf(dmin2, __fmaf_rn((float)(z1b + 1), g.h, g.oz) - Qp.z);
                const float dm2 = dmin2 - margin;
                const float want = lim * powf(1.7f * (float)S / (float)max(cnt - 1, 1), 0.6667f);   // r^2 ~ count^(2/3)
                const float g1 = covered ? INFINITY : dm * dm;               // what the staged cells alone guarantee
                const float lim2 = dmin2 == INFINITY ? want : fminf(dm2 * dm2, want);
                if (!(lim2 > lim)) { retry(q); continue; }
                lim = lim2;
                scan_staged();
                if (lim > g1)   // beyond the first ring's guarantee: the shell cells have to be looked at as well
                for (int zz = z0b; zz <= z1b; ++zz)
                    for (int yy = y0b; yy <= y1b; ++yy) {
                        const int rowbase = (zz * g.ny + yy) * g.nx;
                        const bool inner = zz >= z0 && zz <= z1 && yy >= y0 && yy <= y1;   // this row's cells x0..x1 are staged
                        for (int part = 0; part < (inner ? 2 : 1); ++part) {
                            const int xa = inner ? (part == 0 ? x0b : x1 + 1) : x0b;
                            const int xb = inner ? (part == 0 ? x0 - 1 : x1b) : x1b;
                            if (xa > xb) continue;
                            const int rs = cs[rowbase + xa], re = cs[rowbase + xb + 1];
                            for (int p0 = rs; p0 < re; p0 += 64) {
                                const int pp = p0 + lane;
                                const float4 c = sorted[min(pp, re - 1)];
                                const float d = sqdist3(Qp.x, Qp.y, Qp.z, c.x, c.y, c.z);
                                const bool near = pp < re && d < lim;
                                const unsigned long long mk = __ballot(near);
                                const int pos = cnt + __popcll(mk & lt_mask);
                                if (near && pos < CELL_NEAR) nl[pos] = make_uint2(__float_as_uint(d), (unsigned)(-(pp + 1)));
                                cnt += __popcll(mk);
                            }
                        }
                    }
                if (cnt > CELL_NEAR || cnt <= S) { retry(q); continue; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float e[4]; int ej[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = lane + 64 * r;
                const uint2 t = nl[k];
                const bool v = k < cnt;
                e[r] = v ? __uint_as_float(t.x) : INFINITY;
                ej[r] = v ? (int)t.y : 0;
            }
            // ---- B: exact threshold of rank S + 1 (the query's own distance 0 is rank 1)
            const int S1 = S + 1;
            const float hi = lim;   // every listed distance is < lim (finite: capped by the density radius)
            const float scale = 64.0f / hi;
            hist[lane] = 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            int bin[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bin[r] = min(63, (int)(e[r] * scale));       // monotone in the distance; INF (idle slots) -> 63, not counted
                if (lane + 64 * r < cnt) atomicAdd(&hist[bin[r]], 1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int h = hist[lane];
            const int incl = wave_incl_scan(h, lane);
            const unsigned long long ge = __ballot(incl >= S1);   // != 0: cnt >= S + 1
            const int B = __ffsll((long long)ge) - 1;
            const int nB = rl_i(h, B);
            const int r_need = S1 - (rl_i(incl, B) - nB);         // 1 .. nB elements of bin B belong to the answer
            float tau;
            {
                // the elements of bin B, compacted to lanes 0 .. nB-1 (nB is small: ~2 on uniform clouds)
                int base = 0;
                bool too_many = false;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool inb = (lane + 64 * r < cnt) && bin[r] == B;
                    const unsigned long long mk = __ballot(inb);
                    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, base));
                    if (inb && pos < 64) sd[pos] = e[r];
                    base += __popcll(mk);
                }
                too_many = base > 64;
                if (too_many) { retry(q); continue; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const float v = lane < nB ? sd[lane] : INFINITY;
                int rank = 0;
                for (int u = 0; u < nB; ++u) rank += (rl_f(v, u) < v) ? 1 : 0;
                // tau = the r_need-th smallest of bin B; equal values share a rank, so "<= tau" may select more than r_need
                const float tv = (lane < nB && rank < r_need) ? v : -1.0f;
                tau = wave_max(tv);
                const int n_le = __popcll(__ballot(lane < nB && v <= tau));
                if (n_le != r_need) { tie(q); continue; }         // the cut falls between equal distances
            }
            // ---- C: the S selected elements, one per lane, sorted
            {
                int base = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool sel = e[r] <= tau && ej[r] != qslot;   // idle slots hold +inf; the query itself is not its own neighbour
                    const unsigned long long mk = __ballot(sel);
                    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, base));
                    if (sel && pos < 64) { sd[pos] = e[r]; sj[pos] = ej[r]; }
                    base += __popcll(mk);
                }
                if (base != S) { retry(q); continue; }            // cannot happen (kept as a guard)
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float d = lane < S ? sd[lane] : INFINITY;
            int j = lane < S ? sj[lane] : 0;
            wave_sort64(d, j, lane);
            {
                const float nxd = __shfl_down(d, 1, 64);
                const bool t = (lane + 1 < S && d == nxd) || (lane == 0 && d == 0.f);   // equal neighbours / a twin of the query
                if (__ballot(t) != 0ull) { tie(q); continue; }
            }
            // ---- output: row = [query itself (distance 0), the S neighbours]
            const float4 nb = j >= 0 ? cand[j] : sorted[-j - 1];   // staged slot, or a second-ring point of the sorted array
            const int ni = __float_as_int(nb.w);
            if (lane == 0) {
                if (o.idx) o.idx[(size_t)q * nsample] = q;
                if (o.dist2) o.dist2[(size_t)q * nsample] = 0.f;
            }
            if (lane < S) {
                if (o.idx) o.idx[(size_t)q * nsample + lane + 1] = ni;
                if (o.dist2) o.dist2[(size_t)q * nsample + lane + 1] = d;
                if (o.group_idx) o.group_idx[(size_t)q * S + lane] = ni;
                if (o.ppf) {
                    const float* pn = o.ref_normals + (size_t)ni * 3;
                    const float* qn = o.query_normals + (size_t)q * 3;
                    reinterpret_cast<float4*>(o.ppf)[(size_t)q * S + lane] =
                        ppf4(Qp.x, Qp.y, Qp.z, qn[0], qn[1], qn[2], nb.x, nb.y, nb.z, pn[0], pn[1], pn[2]);
                }
            }
        }
    }
}

// Queries that are not the reference points themselves: counting-sort their indices by the cell of the REFERENCE grid
// they fall into (one workgroup per cloud), so that the lane kernel walks them in cell order too.
template <int CAP>   // cells of the reference grid (the cap it was built with: grid_build_kernel<CAP>)
__global__ __launch_bounds__(1024) void sort_queries_kernel(const float* __restrict__ new_xyz, const int* __restrict__ new_offset,
                                                            const RoitrGrid* __restrict__ grids, int* __restrict__ qorder)
{
    __shared__ int cnt[CAP];
    __shared__ int wave_tot[16];
    const int c = blockIdx.x;
    const int start = c == 0 ? 0 : new_offset[c - 1], end = new_offset[c];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const RoitrGrid g = grids[c];
    const int ncell = g.nx * g.ny * g.nz;
    auto cell_of = [&](int k) {
        const float x = new_xyz[(size_t)k * 3], y = new_xyz[(size_t)k * 3 + 1], z = new_xyz[(size_t)k * 3 + 2];
        const int cx = (int)fminf(fmaxf((x - g.ox) * g.inv_h, 0.f), (float)(g.nx - 1));
        const int cy = (int)fminf(fmaxf((y - g.oy) * g.inv_h, 0.f), (float)(g.ny - 1));
        const int cz = (int)fminf(fmaxf((z - g.oz) * g.inv_h, 0.f), (float)(g.nz - 1));
        return (cz * g.ny + cy) * g.nx + cx;
    };
    for (int k = tid; k < ncell; k += 1024) cnt[k] = 0;
    __syncthreads();
    for (int k = start + tid; k < end; k += 1024) atomicAdd(&cnt[cell_of(k)], 1);
    __syncthreads();
    constexpr int PER = CAP / 1024;
    int loc[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { const int k = tid * PER + j; loc[j] = k < ncell ? cnt[k] : 0; sum += loc[j]; }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; ++w) run += wave_tot[w];
#pragma unroll
    for (int j = 0; j < PER; ++j) { const int k = tid * PER + j; if (k < ncell) cnt[k] = run; run += loc[j]; }
    __syncthreads();
    for (int k = start + tid; k < end; k += 1024) qorder[start + atomicAdd(&cnt[cell_of(k)], 1)] = k;
}

// ---------------------------------------------------------------- lane-per-query kernels: result rows of one lane
template <int L>
__device__ __forceinline__ void lane_emit(const KnnOut& o, int q, int nsample, const float (&d)[L], const int (&id)[L], bool tie,
                                          const float* __restrict__ xyz, float qx, float qy, float qz)
{
    if (tie) {
        const int slot = atomicAdd(o.tie_count, 1);
        o.tie_list[slot] = q;
        return;
    }
#pragma unroll
    for (int j = 0; j < L; ++j) {
        if (j < nsample) {
            if (o.idx) o.idx[(size_t)q * nsample + j] = id[j];
            if (o.dist2) o.dist2[(size_t)q * nsample + j] = d[j];
            if (j >= 1 && o.group_idx) o.group_idx[(size_t)q * (nsample - 1) + j - 1] = id[j];
        }
    }
    if (o.ppf) {  // neighbour indices are read back from the row this lane just wrote
        const float* qn = o.query_normals + (size_t)q * 3;
        const float nx = qn[0], ny = qn[1], nz = qn[2];
        for (int j = 0; j < nsample - 1; ++j) {
            const int gi = o.group_idx[(size_t)q * (nsample - 1) + j];
            const float* pp = xyz + (size_t)gi * 3;
            const float* pn = o.ref_normals + (size_t)gi * 3;
            reinterpret_cast<float4*>(o.ppf)[(size_t)q * (nsample - 1) + j] =
                ppf4(qx, qy, qz, nx, ny, nz, pp[0], pp[1], pp[2], pn[0], pn[1], pn[2]);
        }
    }
}

// ---------------------------------------------------------------- small k, one LANE per query, distance prefilter (round 3)
// The ring-expanding lane kernel below spends its time in the sorted-insertion chain: L compare / select pairs per list slot, and
// the whole wave runs it whenever ANY lane accepts a candidate -- with 64 lists filling at once that is every candidate of the
// 27-cell neighbourhood (17 k / 37 k VALU instructions per wave at L = 10 / 18, SQ pass of round 2).  Here the scan over the
// 3 x 3 x 3 box only FILTERS: a candidate survives if its distance is below tau^2, tau = min(guarantee radius of the box, the
// radius expected to hold ~2 (nsample + 1) points at the box's own density), and its position in the sorted array is appended
// to the lane's column of an LDS table ([slot][thread]: conflict-free).  Everything within the guarantee radius lies inside
// the box, so with at least nsample + 1 survivors the nsample + 1 nearest points of the cloud are among them: the insertion
// chain then runs once per SURVIVOR (~2 (nsample + 1)) instead of once per candidate (13 (nsample + 1)), on lanes that all have
// work.  Same distances, same list, same tie rule -> bit-identical results.  A lane with too few (sparse box, cloud border) or
// too many survivors (CAP) hands its query to the ring-expanding kernel through the retry list.  Blocks take contiguous
// ranges of the cell order per XCD (one L2 serves a neighbourhood: the round-2 launches fetched every cell 4.5 x).
template <int L, int CAP>
__global__ __launch_bounds__(256) void knn_prefilter_kernel(int m, int nsample, int b, const float* __restrict__ xyz,
                                                            const float* __restrict__ new_xyz, const int* __restrict__ offset,
                                                            const int* __restrict__ new_offset, const RoitrGrid* __restrict__ grids,
                                                            const int* __restrict__ cell_start, const float4* __restrict__ sorted, KnnOut o,
                                                            int self_sorted, const int* __restrict__ qorder, int* __restrict__ retry_count,
                                                            int* __restrict__ retry_list)
{
    __shared__ int surv[CAP][256];
    const int nblk = (m + 255) >> 8;
    const int blk = xcd_block_id(nblk);
    if (blk >= nblk) return;
    const int tid = threadIdx.x;
    const int t = blk * 256 + tid;
    if (t >= m) return;
    const int q = self_sorted ? __float_as_int(sorted[t].w) : (qorder ? qorder[t] : t);
    const int seg = segment_of(q, new_offset, b);
    const int start = seg == 0 ? 0 : offset[seg - 1];
    const RoitrGrid g = grids[seg];
    const int* cs = cell_start + (size_t)seg * (GRID_MAX_CELLS + 1);
    const float qx = new_xyz[(size_t)q * 3], qy = new_xyz[(size_t)q * 3 + 1], qz = new_xyz[(size_t)q * 3 + 2];
    int c0[3];
    {
        const float tq[3] = {(qx - g.ox) * g.inv_h, (qy - g.oy) * g.inv_h, (qz - g.oz) * g.inv_h};
        const int dim[3] = {g.nx, g.ny, g.nz};
#pragma unroll
        for (int a = 0; a < 3; ++a) c0[a] = (int)fminf(fmaxf(tq[a], 0.f), (float)(dim[a] - 1));
    }
    const int x0 = max(c0[0] - 1, 0), x1 = min(c0[0] + 1, g.nx - 1);
    const int y0 = max(c0[1] - 1, 0), y1 = min(c0[1] + 1, g.ny - 1);
    const int z0 = max(c0[2] - 1, 0), z1 = min(c0[2] + 1, g.nz - 1);
    // guarantee radius: distance to the nearest face of the box with unseen cells behind it (the ring kernel's stop rule)
    float dmin = INFINITY;
    if (x0 > 0) dmin = fminf(dmin, qx - __fmaf_rn((float)x0, g.h, g.ox));
    if (x1 < g.nx - 1) dmin = fminf(dmin, __fmaf_rn((float)(x1 + 1), g.h, g.ox) - qx);
    if (y0 > 0) dmin = fminf(dmin, qy - __fmaf_rn((float)y0, g.h, g.oy));
    if (y1 < g.ny - 1) dmin = fminf(dmin, __fmaf_rn((float)(y1 + 1), g.h, g.oy) - qy);
    if (z0 > 0) dmin = fminf(dmin, qz - __fmaf_rn((float)z0, g.h, g.oz));
    if (z1 < g.nz - 1) dmin = fminf(dmin, __fmaf_rn((float)(z1 + 1), g.h, g.oz) - qz);
    const float dm = dmin - 2e-4f * g.h;
    // points in the box (from the cell index alone) -> the radius expected to hold ~2 (nsample + 1) of them
    int nbox = 0;
    for (int cz = z0; cz <= z1; ++cz)
        for (int cy = y0; cy <= y1; ++cy) {
            const int rowbase = (cz * g.ny + cy) * g.nx;
            nbox += cs[rowbase + x1 + 1] - cs[rowbase + x0];
        }
    const float vbox = (float)((x1 - x0 + 1) * (y1 - y0 + 1) * (z1 - z0 + 1)) * g.h * g.h * g.h;
    const float want = fmaxf(2.0f * (float)(nsample + 1), (float)(nsample + 9));
    const float r0 = cbrtf(want * vbox * 0.2387324f / (float)max(nbox, 1));   // 3 / (4 pi)
    // near the faces of the cloud's bounding box part of the sphere is empty space: grow the radius by the spherical caps cut
    // off (two fixed-point steps; overlapping caps at edges / corners are subtracted twice -> a larger radius, never a wrong one)
    const float fd[6] = {qx - g.ox, g.ox + (float)g.nx * g.h - qx, qy - g.oy, g.oy + (float)g.ny * g.h - qy, qz - g.oz, g.oz + (float)g.nz * g.h - qz};
    float r_est = r0;
#pragma unroll
    for (int itr = 0; itr < 2; ++itr) {
        float inside = 1.0f;
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const float hc = r_est - fmaxf(fd[f], 0.f);
            if (hc > 0.f) inside -= hc * hc * (3.0f * r_est - hc) / (4.0f * r_est * r_est * r_est);
        }
        r_est = r0 * rcbrtf(fmaxf(inside, 0.125f));
    }
    const float tau = dm > 0.f ? fminf(dm, r_est) : 0.f;
    const float tau2 = tau * tau;
    int cnt = 0;
    for (int cz = z0; cz <= z1; ++cz)
        for (int cy = y0; cy <= y1; ++cy) {
            const int rowbase = (cz * g.ny + cy) * g.nx;
            const int s = cs[rowbase + x0], e = cs[rowbase + x1 + 1];
            constexpr int NF = 8;
            for (int p = s; p < e; p += NF) {
                float4 c[NF];
#pragma unroll
                for (int u = 0; u < NF; ++u) c[u] = sorted[min(p + u, e - 1)];
#pragma unroll
                for (int u = 0; u < NF; ++u) {
                    const float dd = sqdist3(qx, qy, qz, c[u].x, c[u].y, c[u].z);
                    if (p + u < e && dd < tau2) {
                        if (cnt < CAP) surv[cnt][tid] = p + u;
                        ++cnt;
                    }
                }
            }
        }
    if (cnt < nsample + 1 || cnt > CAP) {
        // too few: the (nsample + 1)-th neighbour may lie outside the radius; too many: the table column is full.
        // A box that covers its whole cloud has dm = inf, but tau = min(dm, r_est) = r_est still cuts at the density radius: a
        // query with too few survivors there goes to the ring kernel like any other (a second pass, exact results); that kernel
        // also fills the tail like the reference (KNN_FILL, segment start) when the cloud has fewer than nsample + 1 points.
        const int slot = atomicAdd(retry_count, 1);
        retry_list[slot] = q;
        return;
    }
    float d[L]; int id[L];
#pragma unroll
    for (int j = 0; j < L; ++j) { d[j] = KNN_FILL; id[j] = start; }
    // survivors are re-read from the sorted array (L1 / L2-resident: just scanned) one at a time.  Measured at 512 pairs: fetching
    // them 8 at a time ahead of 8 unrolled insertions is SLOWER (3.65 vs 2.24 ms for the level-1 call): the straight loop lets
    // the 4 resident waves per SIMD cover each other's round trips, the batched form spends the registers on one wave's loads.
    constexpr int NS = 1;
    for (int j0 = 0; j0 < cnt; j0 += NS) {
        float4 c[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) c[u] = sorted[surv[min(j0 + u, cnt - 1)][tid]];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const float dd = j0 + u < cnt ? sqdist3(qx, qy, qz, c[u].x, c[u].y, c[u].z) : INFINITY;
            const int ci = __float_as_int(c[u].w);
            // no `if (dd < d[L - 1])` around the chain (round 4): nearly every survivor is inserted by SOME lane of the wave, so the
            // guard only added a divergent branch whose join made hipcc copy the whole list every step (27 v_mov of ~100
            // instructions per step in the ISA); dd = inf changes nothing
            {
#pragma unroll
                for (int v = L - 1; v > 0; --v) {
                    const bool sh = d[v - 1] > dd, here = d[v] > dd;
                    d[v] = __builtin_amdgcn_fmed3f(dd, d[v - 1], d[v]);   // ascending list: the median IS the shifted / inserted / kept value
                    id[v] = sh ? id[v - 1] : (here ? ci : id[v]);
                }
                const bool h0 = d[0] > dd;
                d[0] = h0 ? dd : d[0]; id[0] = h0 ? ci : id[0];
            }
        }
    }
    bool tie = false;
#pragma unroll
    for (int j = 0; j + 1 < L; ++j) tie |= (j < nsample) && (d[j] == d[j + 1]) && (d[j] < KNN_FILL);
    lane_emit<L>(o, q, nsample, d, id, tie, xyz, qx, qy, qz);
}

// ---------------------------------------------------------------- radius test, one LANE per query (round 5)
// cells [lo, hi] along one axis that the interval [q - rb, q + rb] overlaps (clamped in float first: no int overflow far outside)
__device__ __forceinline__ void ball_cells(float q, float rb, float o, float inv_h, int dim, int& lo, int& hi)
{
    lo = (int)fminf(fmaxf((q - rb - o) * inv_h, 0.f), (float)(dim - 1));
    hi = (int)fminf(fmaxf((q + rb - o) * inv_h, 0.f), (float)(dim - 1));
}

// Nearest-neighbour distance for a radius test (roitr_knn_within: lib/utils.py:509-521 only evaluates `nearest distance < radius`): the
// cells overlapping the bounding box of the ball of radius sqrt(cap2) hold every point closer than that, so the minimum over them
// is the exact nearest distance whenever it is below cap2, and some value >= cap2 (1e10 when the box is empty) otherwise.  The ring
// kernel it replaces (L = 2) scanned the query's cell and, for the 3 / 4 of the queries closer to a face of their cell than the
// radius, the 26 cells around it.
__global__ __launch_bounds__(256) void knn_within_kernel(int m, int b, const float* __restrict__ new_xyz, const int* __restrict__ new_offset,
                                                         const RoitrGrid* __restrict__ grids, const int* __restrict__ cell_start,
                                                         const float4* __restrict__ sorted, const int* __restrict__ qorder, float cap2,
                                                         float* __restrict__ dist2)
{
    const int nblk = (m + 255) >> 8;
    const int blk = xcd_block_id(nblk);
    if (blk >= nblk) return;
    const int t = blk * 256 + threadIdx.x;
    if (t >= m) return;
    const int q = qorder ? qorder[t] : t;
    const int seg = segment_of(q, new_offset, b);
    const RoitrGrid g = grids[seg];
    const int* cs = cell_start + (size_t)seg * (GRID_MAX_CELLS + 1);
    const float qx = new_xyz[(size_t)q * 3], qy = new_xyz[(size_t)q * 3 + 1], qz = new_xyz[(size_t)q * 3 + 2];
    const float rb = sqrtf(cap2) * 1.0001f + 2e-4f * g.h;
    int bx0, bx1, by0, by1, bz0, bz1;
    ball_cells(qx, rb, g.ox, g.inv_h, g.nx, bx0, bx1);
    ball_cells(qy, rb, g.oy, g.inv_h, g.ny, by0, by1);
    ball_cells(qz, rb, g.oz, g.inv_h, g.nz, bz0, bz1);
    // a query farther than the radius from the grid's box has nothing within the radius: the clamped cells are not scanned
    const bool outside = qx + rb < g.ox || qx - rb > g.ox + (float)g.nx * g.h || qy + rb < g.oy || qy - rb > g.oy + (float)g.ny * g.h ||
                         qz + rb < g.oz || qz - rb > g.oz + (float)g.nz * g.h;
    float best = KNN_FILL;
    if (!outside)
    for (int cz = bz0; cz <= bz1; ++cz)
        for (int cy = by0; cy <= by1; ++cy) {
            const int rowbase = (cz * g.ny + cy) * g.nx;
            const int s = cs[rowbase + bx0], e = cs[rowbase + bx1 + 1];
            constexpr int NF = 4;
            for (int p = s; p < e; p += NF) {
                float4 c[NF];
#pragma unroll
                for (int u = 0; u < NF; ++u) c[u] = sorted[min(p + u, e - 1)];   // a repeated last candidate changes no minimum
#pragma unroll
                for (int u = 0; u < NF; ++u) best = fminf(best, sqdist3(qx, qy, qz, c[u].x, c[u].y, c[u].z));
            }
        }
    dist2[q] = best;
}

// ---------------------------------------------------------------- grid query, one LANE per query
// For large query counts (tens of thousands and up: every level-1/2 call of a multi-pair batch) a wave per query
// spends ~40x more instructions than the arithmetic needs.  Here every lane owns a query and its own sorted list in
// registers (L = template capacity >= nsample+1); lanes walk the same Chebyshev rings over the same grid.  Self
// queries are taken in CELL ORDER (the counting-sorted order), so the 64 lanes of a wave search neighbouring cells:
// their trip counts agree and their candidate loads hit the same cache lines.  Same exactness rule as the wave
// kernel: ties among the best nsample+1 distances -> the query goes to the replay kernel.
template <int L>
__global__ __launch_bounds__(256) void knn_lane_kernel(int m, int nsample, int b, const float* __restrict__ xyz,
                                                       const float* __restrict__ new_xyz, const int* __restrict__ offset,
                                                       const int* __restrict__ new_offset, const RoitrGrid* __restrict__ grids,
                                                       const int* __restrict__ cell_start, const float4* __restrict__ sorted, KnnOut o,
                                                       int self_sorted, const int* __restrict__ qorder, float cap2,
                                                       const int* __restrict__ list, const int* __restrict__ list_count)
{
  // list mode (list != nullptr): the queries knn_prefilter_kernel could not decide, *list_count of them, grid-stride.
  // Otherwise one query per thread, launched with xcd_grid(blocks): every XCD takes a contiguous eighth of the cell order, so a
  // cloud's sorted points are fetched into ONE L2 (round 5; blocks b, b + 1, ... of the plain order land on 8 different XCDs and
  // every L2 pulled every cloud).
  const int total = list ? *list_count : m;
  const int nblk_ = (m + 255) >> 8;
  const int blk0 = list ? (int)blockIdx.x : xcd_block_id(nblk_);
  if (!list && blk0 >= nblk_) return;
  for (int t = blk0 * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
    const int q = list ? list[t] : (self_sorted ? __float_as_int(sorted[t].w) : (qorder ? qorder[t] : t));
    int lo = 0, hi = b - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (q < new_offset[mid]) hi = mid; else lo = mid + 1; }
    const int seg = lo;
    const int start = seg == 0 ? 0 : offset[seg - 1];
    const RoitrGrid g = grids[seg];
    const int* cs = cell_start + (size_t)seg * (GRID_MAX_CELLS + 1);
    const float qx = new_xyz[(size_t)q * 3], qy = new_xyz[(size_t)q * 3 + 1], qz = new_xyz[(size_t)q * 3 + 2];
    float d[L]; int id[L];
#pragma unroll
    for (int j = 0; j < L; ++j) { d[j] = KNN_FILL; id[j] = start; }
    int c0[3];
    {
        const float tq[3] = {(qx - g.ox) * g.inv_h, (qy - g.oy) * g.inv_h, (qz - g.oz) * g.inv_h};
        const int dim[3] = {g.nx, g.ny, g.nz};
#pragma unroll
        for (int a = 0; a < 3; ++a) c0[a] = (int)fminf(fmaxf(tq[a], 0.f), (float)(dim[a] - 1));
    }
    auto offer = [&](float dd, int ci) {
        if (dd < d[L - 1]) {
#pragma unroll
            for (int j = L - 1; j > 0; --j) {
                const bool sh = d[j - 1] > dd, here = d[j] > dd;
                d[j] = __builtin_amdgcn_fmed3f(dd, d[j - 1], d[j]);   // ascending list: the median IS the shifted / inserted / kept value
                id[j] = sh ? id[j - 1] : (here ? ci : id[j]);
            }
            const bool h0 = d[0] > dd;
            d[0] = h0 ? dd : d[0]; id[0] = h0 ? ci : id[0];
        }
    };
    auto scan = [&](int s, int e) {
        // NF candidate loads in flight per lane (clamped addresses, masked distances): the loop is otherwise one
        // dependent L2 round trip per candidate
        constexpr int NF = 8;
        for (int p = s; p < e; p += NF) {
            float4 c[NF];
#pragma unroll
            for (int u = 0; u < NF; ++u) c[u] = sorted[min(p + u, e - 1)];
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                const float dd = p + u < e ? sqdist3(qx, qy, qz, c[u].x, c[u].y, c[u].z) : INFINITY;
                offer(dd, __float_as_int(c[u].w));
            }
        }
    };
    const float margin = 2e-4f * g.h;
    const int maxr = max(max(g.nx, g.ny), g.nz);
    for (int r = 0; r <= maxr; ++r) {
        const int x0 = max(c0[0] - r, 0), x1 = min(c0[0] + r, g.nx - 1);
        const int y0 = max(c0[1] - r, 0), y1 = min(c0[1] + r, g.ny - 1);
        const int z0 = max(c0[2] - r, 0), z1 = min(c0[2] + r, g.nz - 1);
        for (int cz = z0; cz <= z1; ++cz) {
            for (int cy = y0; cy <= y1; ++cy) {
                const int rowbase = (cz * g.ny + cy) * g.nx;
                const bool face = (r == 0) || (cy == c0[1] - r) || (cy == c0[1] + r) || (cz == c0[2] - r) || (cz == c0[2] + r);
                // a row of cells whose (y, z) slab lies farther from the query than the current worst list entry (or than the cap
                // of roitr_knn_within) cannot contribute: nothing in it passes `dd < d[L - 1]` (round 3; same rounding margin as
                // the stop rule below).  Lanes of a wave walk neighbouring cells, so they mostly agree.  (Trimming the cells of a face
                // row from both ends by the same bound costs more than it saves: kNN 16.5 vs 11.5 ms per 512-pair step.)
                if (r > 0) {
                    const float ylo = __fmaf_rn((float)cy, g.h, g.oy), zlo = __fmaf_rn((float)cz, g.h, g.oz);
                    const float dy = fmaxf(fmaxf(ylo - qy, qy - (ylo + g.h)), 0.f), dz = fmaxf(fmaxf(zlo - qz, qz - (zlo + g.h)), 0.f);
                    const float lb = fmaxf(sqrtf(dy * dy + dz * dz) - margin, 0.f);
                    if (lb * lb > fminf(d[L - 1], cap2)) continue;
                }
                if (face) scan(cs[rowbase + x0], cs[rowbase + x1 + 1]);
                else {
                    const int xa = c0[0] - r, xb = c0[0] + r;
                    if (xa >= 0) scan(cs[rowbase + xa], cs[rowbase + xa + 1]);
                    if (xb < g.nx) scan(cs[rowbase + xb], cs[rowbase + xb + 1]);
                }
            }
        }
        float dmin = INFINITY;
        if (x0 > 0) dmin = fminf(dmin, qx - __fmaf_rn((float)x0, g.h, g.ox));
        if (x1 < g.nx - 1) dmin = fminf(dmin, __fmaf_rn((float)(x1 + 1), g.h, g.ox) - qx);
        if (y0 > 0) dmin = fminf(dmin, qy - __fmaf_rn((float)y0, g.h, g.oy));
        if (y1 < g.ny - 1) dmin = fminf(dmin, __fmaf_rn((float)(y1 + 1), g.h, g.oy) - qy);
        if (z0 > 0) dmin = fminf(dmin, qz - __fmaf_rn((float)z0, g.h, g.oz));
        if (z1 < g.nz - 1) dmin = fminf(dmin, __fmaf_rn((float)(z1 + 1), g.h, g.oz) - qz);
        if (dmin == INFINITY) break;
        float tau = d[L - 1];  // the (nsample+1)-th best must be final before stopping
#pragma unroll
        for (int j = 0; j < L - 1; ++j) tau = (j == nsample) ? d[j] : tau;
        const float dm = dmin - margin;
        // cap2 < inf (roitr_knn_within): the caller only compares the distances against sqrt(cap2), so the search may stop
        // as soon as everything unseen is farther than that
        if (dm > 0.f && fminf(tau, cap2) < dm * dm) break;
    }
    bool tie = false;
#pragma unroll
    for (int j = 0; j + 1 < L; ++j) tie |= (j < nsample) && (d[j] == d[j + 1]) && (d[j] < KNN_FILL);
    lane_emit<L>(o, q, nsample, d, id, tie, xyz, qx, qy, qz);
  }
}

// ---------------------------------------------------------------- lane-per-query, small clouds (no grid)
// Clouds below the grid threshold (the two coarse levels: 312 and 78 points per cloud at N = 5000) used to go to the
// wave-per-query brute kernel -- 80 k waves for 80 k queries, 0.29 ms per call.  Here a lane owns a query and keeps its
// sorted best list in registers (same list code and same tie -> replay rule as knn_lane_kernel); a block (one wave) stages the
// cloud(s) its 64 consecutive queries belong to through LDS as float4 (x, y, z, index) and every lane reads the same
// candidate per step (LDS broadcast).
template <int L>
__global__ __launch_bounds__(64) void knn_lane_brute_kernel(int m, int nsample, int b, const float* __restrict__ xyz,
                                                             const float* __restrict__ new_xyz, const int* __restrict__ offset,
                                                             const int* __restrict__ new_offset, KnnOut o, int CH)
{
    // CH candidates per staging chunk: dynamic LDS sized by the launcher from the mean cloud (16 CH bytes; a 312-point cloud needs 5 KB,
    // not the 16 KB of a fixed 1024 -- one wave per workgroup: the LDS decides how many of them a CU holds, and what is left for
    // the feature path beside them)
    extern __shared__ __attribute__((aligned(16))) float4 cand[];
    __shared__ int seg_range[2];
    const int tid = threadIdx.x;
    const int q = blockIdx.x * 64 + tid;   // one wave per block: many small blocks interleave their insertion chains
    const bool valid = q < m;
    int seg = 0;
    if (valid) {
        if (b > 0) { int lo = 0, hi = b - 1; while (lo < hi) { const int mid = (lo + hi) >> 1; if (q < new_offset[mid]) hi = mid; else lo = mid + 1; } seg = lo; }
        else { while (!(q < new_offset[seg])) seg++; }
    }
    if (tid == 0) seg_range[0] = seg;
    if (q == min(m, (int)(blockIdx.x + 1) * 64) - 1) seg_range[1] = seg;
    __syncthreads();
    const int s_lo = seg_range[0], s_hi = seg_range[1];
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (valid) { qx = new_xyz[(size_t)q * 3]; qy = new_xyz[(size_t)q * 3 + 1]; qz = new_xyz[(size_t)q * 3 + 2]; }
    const int my_start = seg == 0 ? 0 : offset[seg - 1];
    float d[L]; int id[L];
#pragma unroll
    for (int j = 0; j < L; ++j) { d[j] = KNN_FILL; id[j] = my_start; }
    auto offer = [&](float dd, int ci) {
        if (dd < d[L - 1]) {
#pragma unroll
            for (int j = L - 1; j > 0; --j) {
                const bool sh = d[j - 1] > dd, here = d[j] > dd;
                d[j] = __builtin_amdgcn_fmed3f(dd, d[j - 1], d[j]);   // ascending list: the median IS the shifted / inserted / kept value
                id[j] = sh ? id[j - 1] : (here ? ci : id[j]);
            }
            const bool h0 = d[0] > dd;
            d[0] = h0 ? dd : d[0]; id[0] = h0 ? ci : id[0];
        }
    };
    for (int s = s_lo; s <= s_hi; ++s) {
        const int start = s == 0 ? 0 : offset[s - 1], end = offset[s];
        for (int base = start; base < end; base += CH) {
            const int cnt = min(CH, end - base);
            __syncthreads();
            for (int i = tid; i < cnt; i += 64) {
                const float* pp = xyz + (size_t)(base + i) * 3;
                cand[i] = make_float4(pp[0], pp[1], pp[2], __int_as_float(base + i));
            }
            __syncthreads();
            if (valid && seg == s) {
                for (int i = 0; i < cnt; ++i) {
                    const float4 c = cand[i];
                    offer(sqdist3(qx, qy, qz, c.x, c.y, c.z), __float_as_int(c.w));
                }
            }
        }
    }
    if (!valid) return;
    bool tie = false;
#pragma unroll
    for (int j = 0; j + 1 < L; ++j) tie |= (j < nsample) && (d[j] == d[j + 1]) && (d[j] < KNN_FILL);
    lane_emit<L>(o, q, nsample, d, id, tie, xyz, qx, qy, qz);
}

// ---------------------------------------------------------------- exact replay of tied queries
// knnquery_cuda_kernel.cu:65-108 restated for one wave: the heap lives in LDS, every lane runs the
// same (uniform) heap code; 64 distances are evaluated per step and only those below the root are
// offered to the heap, in index order -- the heap sees exactly the reference's insertion sequence.
// A replayed query admits ~k ln(n / k) candidates one after the other and that serial chain (~2 us per admission) is all
// its run time: 0.5 ms at k = 64, n = 30000 (unchanged by keeping the sinking value in registers and dropping the block
// barriers, the form below: 2.94 vs 2.99 ms for the config-5 call).  Measured and dropped: the heap in registers (entry p in lane p % 64, reads by
// v_readlane, writes by lane compare + select: no LDS, no barrier), both as a straight restatement (1.46 vs 1.08 ms per launch
// on that case) and with every node's larger child precomputed by four ds_bpermute and a scalar walk down that path (3.10 vs
// 2.91 ms for the whole config-5 call, 0.49 vs 0.39 ms of replay per 32-pair 4DMatch forward): with ONE wave per CU nothing
// hides the SGPR <-> VALU hand-offs of the readlane chain, which cost as much as the LDS round trips they replace; 8 batches of
// candidate loads in flight (kept) change nothing.  One query per block (up to 1024 blocks) keeps the kernel at one chain.
// One wave per block: LDS operations of a wave complete in order, so cross-lane hand-offs only need the counter wait.
__device__ __forceinline__ void lds_sync_wave() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// reheap (knnquery_cuda_kernel.cu:21-36) with the sinking value kept in registers: the entry at `rt` is always the new one; the
// right child wins only when strictly larger, the walk stops at the first child strictly smaller than the value.
// Round 6 (temporary s_memtime counters inside the kernel: a replayed query of the config-5 call spent 1.35 M cycles --
// 461 admissions at 1 840 cycles each = 300 per heap level, 365 k in the candidate loads): the per-level cost was not the LDS
// round trip but the VALU -> SALU hand-offs of a loop whose every decision was a scalar branch or an exec-mask update.  Now
//  * an entry is one 8-byte (distance, index) word, node j lives in slot j + 1: the two children of a node are ONE aligned 16-byte
//    read, the parent's update one 8-byte write nobody waits for, the new root comes back in a register;
//  * the walk is branch-free with a uniform trip count (the heap's depth): a missing child reads as distance -1 (distances are
//    >= 0: it neither wins nor is moved), a walk that has stopped keeps comparing +inf (stops again), and its parent writes go
//    to the unused slot 0 -- every decision is a v_cmp feeding v_cndmask, no exec-mask update and no branch inside the walk.
// Returns the distance now at the root.
constexpr int REPLAY_SLOTS = 208;   // nsample <= 100 nodes in slots 1 .. 100; a stopped walk may read the "children" of a leaf
__device__ __forceinline__ int heap_levels(int k) { return 31 - __builtin_clz(k); }   // depth of node k - 1 = floor(log2 k), k >= 1
__device__ __forceinline__ float sift_down(uint2* hp, int k, int levels, float nd, int ni)
{
    int rt = 0;
    float root = nd, ndc = nd;
    for (int lv = 0; lv < levels; ++lv) {   // wave-uniform trip count
        const int child = 2 * rt + 1;
        // both children, distances AND indices, in one round trip (left to itself hipcc fetches the indices in a second, dependent
        // read behind the comparison); the wait is part of the statement: inline asm is invisible to the waitcnt insertion
        uint4 cc;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(cc) : "v"((unsigned)(uintptr_t)(hp + child + 1)) : "memory");
        const float c0 = child < k ? __uint_as_float(cc.x) : -1.f, c1 = child + 1 < k ? __uint_as_float(cc.z) : -1.f;
        const bool right = c1 > c0;
        const float cd = right ? c1 : c0;
        const unsigned ci = right ? cc.w : cc.y;
        const bool move = !(ndc > cd);          // reheap swaps on equality (l.29)
        hp[move ? rt + 1 : 0] = make_uint2(__float_as_uint(cd), ci);
        if (lv == 0) root = move ? cd : nd;
        rt = move ? child + (right ? 1 : 0) : rt;
        ndc = move ? ndc : INFINITY;
    }
    hp[rt + 1] = make_uint2(__float_as_uint(nd), (unsigned)ni);
    return root;
}

// Round 6, second step: the admissions of a query form a PIPELINE across four lanes.  A sift-down touches every depth of the heap
// once, top to bottom; the next admission only needs (a) the new root -- known after the FIRST level of the one before -- to be
// decided, and (b) at each depth the values the one before left there.  So admission j + 1 may start two levels behind admission
// j: lane a holds the walk of the a-th admission in flight (state: node, value, index), one `step` advances all of them by one
// level with ONE 16-byte LDS read per lane, and a new walk enters every other step -- 2 levels of latency per admission instead
// of 6 - 7.  Order of effects = the sequential reheap: walk j writes depth d (the child moving up, or its own value settling)
// at its step d, walk j + 1 reads depth d at its step d - 1, two steps after walk j started = one step AFTER that write (LDS
// operations of a wave complete in order).  Slots of missing nodes hold distance -1: a walk that runs off the heap stops there
// (-1 neither wins a comparison nor is moved) and settles one step later, still in time.  Bit-exact by construction; the whole
// tie suite (lattice, duplicate, plane clouds: every query tied) is the test.
__global__ __launch_bounds__(64) void knn_replay_kernel(int nsample, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                        const int* __restrict__ offset, const int* __restrict__ new_offset, KnnOut o)
{
    __shared__ __attribute__((aligned(16))) uint2 hp[REPLAY_SLOTS];
    const int lane = threadIdx.x;
    const int count = *o.tie_count;
    for (int t = blockIdx.x; t < count; t += gridDim.x) {
        const int q = o.tie_list[t];
        int start, end, seg;
        find_segment(q, offset, new_offset, start, end, seg, o.b);
        Query Q = {new_xyz[(size_t)q * 3], new_xyz[(size_t)q * 3 + 1], new_xyz[(size_t)q * 3 + 2]};
        for (int p = lane; p < REPLAY_SLOTS; p += 64)
            hp[p] = (p >= 1 && p <= nsample) ? make_uint2(__float_as_uint(KNN_FILL), (unsigned)start) : make_uint2(__float_as_uint(-1.f), 0u);
        __syncthreads();
        float root = KNN_FILL;
        // walk state of this lane (lanes 0 .. 3 carry walks, the others idle along)
        bool act = false; int rt = 0; float w_nd = 0.f, w_cmp = INFINITY; unsigned w_ni = 0u;
        int slot = 0; bool last_started = false;
        const unsigned hp_addr = (unsigned)(uintptr_t)hp;
        // one level for every walk in flight; optionally a new walk (snd, sni) enters at the root in lane `slot`
        auto step = [&](bool start_walk, float snd, int sni) {
            if (start_walk && lane == slot) { act = true; rt = 0; w_nd = snd; w_cmp = snd; w_ni = (unsigned)sni; }
            uint4 cc;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(cc) : "v"(hp_addr + (unsigned)(2 * rt + 2) * 8u) : "memory");
            const float c0 = __uint_as_float(cc.x), c1 = __uint_as_float(cc.z);
            const bool right = c1 > c0;                       // the right child wins only when strictly larger (l.27)
            const float cd = right ? c1 : c0;
            const unsigned ci = right ? cc.w : cc.y;
            const bool move = !(w_cmp > cd);                  // reheap swaps on equality (l.29); an idle lane compares +inf: never
            const uint2 wv = move ? make_uint2(__float_as_uint(cd), ci) : make_uint2(__float_as_uint(w_nd), w_ni);
            hp[act ? rt + 1 : 0] = wv;                        // the child moves up, or the walk's own value settles here
            if (start_walk) { root = rl_f(move ? cd : snd, slot); slot = (slot + 1) & 3; }   // the new root is final after a walk's first level
            rt = move ? 2 * rt + 1 + (right ? 1 : 0) : rt;
            act = act && move;
            w_cmp = move ? w_cmp : INFINITY;
            last_started = start_walk;
        };
        // the scan is one wave walking the whole cloud: NB batches of 64 distances are loaded together (the admission order
        // below is still strictly the index order), otherwise every step is a dependent HBM / L2 round trip
        constexpr int NB = 8;
        // coordinates of the NEXT 512 candidates are requested before the admission chain of the current ones starts (24 loads in
        // flight under ~10 k cycles of heap work): the probe showed 8 dependent round trips per 512 candidates, 27 % of the kernel
        float px[NB], py[NB], pz[NB];
        auto request = [&](int b0) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int k = min(b0 + 64 * u + lane, end - 1);
                const float* p = xyz + (size_t)k * 3;
                px[u] = p[0]; py[u] = p[1]; pz[u] = p[2];
            }
        };
        request(start);
        for (int base0 = start; base0 < end; base0 += 64 * NB) {
            float cdv[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) cdv[u] = base0 + 64 * u + lane < end ? sqdist3(Q.x, Q.y, Q.z, px[u], py[u], pz[u]) : INFINITY;
            if (base0 + 64 * NB < end) request(base0 + 64 * NB);   // wave-uniform
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int base = base0 + 64 * u;
                if (base >= end) break;   // wave-uniform
                const float cd = cdv[u];
                unsigned long long mk = __ballot(cd < root);
                while (mk) {
                    const int l = __ffsll((long long)mk) - 1;
                    mk &= mk - 1;
                    const float nd = rl_f(cd, l);
                    if (nd < root) {   // strict admission (l.97); the new point replaces the root, then reheap (l.21-36)
                        if (last_started) step(false, 0.f, 0);   // a walk enters two levels behind the one before
                        step(true, nd, base + l);
                    }
                }
            }
        }
        for (int d = 0; d < 8; ++d) step(false, 0.f, 0);   // drain: a walk settles after at most depth + 1 <= 7 levels
        lds_sync_wave();
        if (lane == 0) {  // heap_sort, l.39-48: the root goes to slot i, the old slot-i entry sinks from the root over the first i
            for (int i = nsample - 1; i > 0; i--) {
                const uint2 top = hp[1], last = hp[i + 1];
                hp[i + 1] = top;
                (void)sift_down(hp, i, heap_levels(i), __uint_as_float(last.x), (int)last.y);
            }
        }
        lds_sync_wave();
        float d[2]; int i[2];
        const uint2 e0 = hp[(lane < nsample ? lane : 0) + 1], e1 = hp[(lane + 64 < nsample ? lane + 64 : 0) + 1];
        d[0] = lane < nsample ? __uint_as_float(e0.x) : 0.f; i[0] = lane < nsample ? (int)e0.y : 0;
        d[1] = lane + 64 < nsample ? __uint_as_float(e1.x) : 0.f; i[1] = lane + 64 < nsample ? (int)e1.y : 0;
        write_rows<2>(o, q, nsample, d, i, lane, xyz, Q);
        __syncthreads();
    }
}

// scratch for the legacy (workspace-less) entry point
int* g_legacy_ws = nullptr;
size_t g_legacy_ws_ints = 0;

}  // namespace

// Workspace layout (ints): [0] tie counter, [1] retry counter (cell kernel -> general kernel), [4 .. 4 + m) tie list, then (grid
// path) grids (b * 8 words), cell_start (b * (GRID_MAX_CELLS+1)), sorted float4 (n * 4 words, 16-B aligned), then m ints: the
// query order of non-self lane queries / the retry list of self queries (never both in one call).
extern "C" size_t roitr_knn_workspace_bytes(int b, int n, int m)
{
    size_t ints = 4 + (size_t)m;
    ints = (ints + 3) & ~(size_t)3;
    ints += (size_t)b * 8 + (size_t)b * (GRID_MAX_CELLS + 1);
    ints = (ints + 3) & ~(size_t)3;
    ints += (size_t)n * 4;
    ints += (size_t)m;  // query order (lane kernel, non-self queries)
    ints += (size_t)m;  // retry list (queries the prefilter kernel hands to the ring-expanding one)
    return ints * 4 + 64;
}

namespace {
struct WsView {
    int* tie_count; int* tie_list; RoitrGrid* grids; int* cell_start; float4* sorted; int* qorder; int* retry;
};
WsView carve(void* ws, int b, int n, int m)
{
    uintptr_t base = ((uintptr_t)ws + 15) & ~(uintptr_t)15;
    int* p = (int*)base;
    WsView v;
    v.tie_count = p; v.tie_list = p + 4;
    size_t ints = 4 + (size_t)m; ints = (ints + 3) & ~(size_t)3;
    v.grids = (RoitrGrid*)(p + ints); ints += (size_t)b * 8;
    v.cell_start = p + ints; ints += (size_t)b * (GRID_MAX_CELLS + 1);
    ints = (ints + 3) & ~(size_t)3;
    v.sorted = (float4*)(p + ints);
    ints += (size_t)n * 4;
    v.qorder = (b > 0 && n > 0) ? p + ints : nullptr;
    v.retry = v.qorder ? v.qorder + m : nullptr;
    return v;
}
}  // namespace

// Builds the per-cloud uniform grids for `xyz` (b clouds, n points) into `ws`; reusable by any number
// of roitr_knnquery_grid calls on the same reference cloud set (same b, n, m-capacity carve).
extern "C" int roitr_knn_build_grid_ex(int b, int n, int m_capacity, const float* xyz, const int* offset, void* ws, float target_occupancy,
                                       hipStream_t stream);
extern "C" int roitr_knn_build_grid(int b, int n, int m_capacity, const float* xyz, const int* offset, void* ws, hipStream_t stream)
{
    return roitr_knn_build_grid_ex(b, n, m_capacity, xyz, offset, ws, 0.f, stream);
}

// target_occupancy: mean points per grid cell the cell size is chosen for (<= 0: default).  The ring-1 neighbourhood
// (27 cells) should hold the k+1 nearest neighbours of almost every query: ~ (k+1)/3 points per cell.
extern "C" int roitr_knn_build_grid_ex(int b, int n, int m_capacity, const float* xyz, const int* offset, void* ws, float target_occupancy,
                                       hipStream_t stream)
{
    if (b <= 0 || n <= 0) return ROITR_OK;
    WsView v = carve(ws, b, n, m_capacity);
    const float rho = target_occupancy > 0.f ? target_occupancy : 6.0f;
    roitr_prof_begin(ROITR_PROF_GRID, 12.0 * n + 16.0 * n, stream);
    if ((long)n <= 6144L * b) grid_build_kernel<4096><<<b, 1024, 0, stream>>>(xyz, offset, v.grids, v.cell_start, v.sorted, rho);
    else grid_build_kernel<GRID_MAX_CELLS><<<b, 1024, 0, stream>>>(xyz, offset, v.grids, v.cell_start, v.sorted, rho);
    roitr_prof_end(ROITR_PROF_GRID, stream);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" const void* roitr_knn_sorted_points(int b, int n, int m_capacity, void* ws)
{
    return carve(ws, b, n, m_capacity).sorted;
}

// The general entry point.  use_grid != 0 requires a prior roitr_knn_build_grid on the same ws.
// Outputs idx / dist2 / group_idx / ppf are each optional (null = not wanted).
namespace {
// cap2: the lane kernels may stop a query's ring search once everything unseen is beyond sqrt(cap2) (roitr_knn_within);
// INFINITY = the exact k nearest neighbours
int knnquery_impl(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz, const int* offset,
                  const int* new_offset, int* idx, float* dist2, int* group_idx, float* ppf,
                  const float* ref_normals, const float* query_normals, int use_grid, int m_capacity, void* ws,
                  float cap2, hipStream_t stream)
{
    if (m <= 0) return ROITR_OK;
    if (nsample < 1 || nsample > 100) return ROITR_ERR_ARG;  // best_dist[100], knnquery_cuda_kernel.cu:86
    if (ppf && (!ref_normals || !query_normals)) return ROITR_ERR_ARG;
    WsView v = carve(ws, b, n, m_capacity);
    KnnOut o = {idx, dist2, group_idx, ppf, ref_normals, query_normals, v.tie_count, v.tie_list, b > 0 ? b : 0};
    ROITR_HIP(hipMemsetAsync(v.tie_count, 0, 2 * sizeof(int), stream));   // tie counter + retry counter
    const int blocks = div_up(m, 4);
    // algorithmic bytes: refs xyz(+normals) once, queries when distinct, idx + dist2/ppf rows out
    {
        const int kk = group_idx || ppf ? nsample - 1 : nsample;
        double bytes = (ppf ? 24.0 : 12.0) * n + (new_xyz != xyz ? (ppf ? 24.0 : 12.0) * m : 0.0);
        bytes += ((idx ? 4.0 * nsample : 0.0) + (dist2 ? 4.0 * nsample : 0.0) + (group_idx ? 4.0 * kk : 0.0) + (ppf ? 16.0 * kk : 0.0)) * m;
        roitr_prof_begin(ROITR_PROF_KNN, bytes, stream);
    }
    // kernel choice by shape only (never by the data): a lane per query from 8192 queries on a grid;
    // small clouds without a grid: a lane per query once there are enough queries to fill the chip that way; below that a wave
    // per query (one pair per call: 624 queries at level 3 were 10 waves scanning 1250 references each, 152 us per call)
    constexpr int lane_min = 8192, lane_brute_min = 65536;
    const bool lane_ok = use_grid && m >= lane_min && (!ppf || group_idx) && b > 0;
    const int self_sorted = (new_xyz == xyz && new_offset == offset && m == n) ? 1 : 0;
    // non-self queries: walk them in reference-cell order; the order array reuses the tie list's tail
    // (tie slots are handed out from the front; a query is either listed as a tie or not, so m slots suffice for both
    //  only if disjoint -> the order array lives in its own region carved behind the sorted points)
    int* qorder = nullptr;
    if (lane_ok && !self_sorted && v.qorder) {
        qorder = v.qorder;
        if ((long)n <= 6144L * b) sort_queries_kernel<4096><<<b, 1024, 0, stream>>>(new_xyz, new_offset, v.grids, qorder);   // the rule of roitr_knn_build_grid_ex
        else sort_queries_kernel<GRID_MAX_CELLS><<<b, 1024, 0, stream>>>(new_xyz, new_offset, v.grids, qorder);
    }
#define LANE_CASE(LC)                                                                                                        \
    knn_lane_kernel<LC><<<xcd_grid(div_up(m, 256)), 256, 0, stream>>>(m, nsample, b, xyz, new_xyz, offset, new_offset, v.grids, v.cell_start, \
                                                             v.sorted, o, self_sorted, qorder, cap2, nullptr, nullptr)
    // prefilter kernel + the ring-expanding kernel in list mode for what it hands over (a fixed grid-stride launch: the count
    // lives on the device)
#define PREF_CASE(LC, CAPC)                                                                                                  \
    do {                                                                                                                     \
        int* retry_count = v.tie_count + 1;                                                                                  \
        knn_prefilter_kernel<LC, CAPC><<<xcd_grid(div_up(m, 256)), 256, 0, stream>>>(m, nsample, b, xyz, new_xyz, offset, new_offset, v.grids, \
                                                                                    v.cell_start, v.sorted, o, self_sorted, qorder, retry_count, v.retry); \
        knn_lane_kernel<LC><<<min(div_up(m, 256), 1024), 256, 0, stream>>>(m, nsample, b, xyz, new_xyz, offset, new_offset, v.grids, v.cell_start, \
                                                                           v.sorted, o, self_sorted, qorder, cap2, v.retry, retry_count);   \
    } while (0)
    if (lane_ok && nsample + 1 <= 34) {   // from nsample + 1 = 35 the selection kernels take over
        const int need = nsample + 1;
        const bool exact_k = cap2 == INFINITY && v.retry != nullptr;   // roitr_knn_within (cap2 < inf) stops on the radius: ring kernel
        // Measured per call at 512 pairs (round 3, rocprofv3 kernel trace): the prefilter form wins where the insertion chain
        // dominated and the grid is fine enough for the radius rule -- the level-1 self query (need 10, 6 points per cell: 3.56 ->
        // 2.24 ms) -- and loses on the 3-NN interpolation queries (need 4: the 4-deep chain was never the cost, 0.84 -> 4.0 ms)
        // and, with grids of 6 points per cell, on need 18 (the sphere of one cell size holds 25 points: most lanes would hand
        // their query over; with 9 per cell 3 x 1.0 -> 3 x (2.1 + 0.45) ms; re-measured at the end of round 3 with a 48 / 56-slot table and
        // target counts of 1.4 - 2.0 (nsample + 1): 8 - 15 % of the need-18 queries still go to the ring kernel -- the clouds are
        // surface-like, the count inside a radius does not follow the box's volume density -- kNN 12.7 -> 17.4 - 19.9 ms per step).
        // So: need 5..10 only.
        if (cap2 < INFINITY && nsample == 1 && dist2 && !idx && !group_idx)   // roitr_knn_within: one pass over the ball's cells
            knn_within_kernel<<<xcd_grid(div_up(m, 256)), 256, 0, stream>>>(m, b, new_xyz, new_offset, v.grids, v.cell_start, v.sorted,
                                                                           self_sorted ? nullptr : qorder, cap2, dist2);
        else if (need <= 2) LANE_CASE(2);
        else if (need <= 4) LANE_CASE(4);
        else if (need <= 10) { if (exact_k) PREF_CASE(10, 40); else LANE_CASE(10); }
        else if (need <= 18) LANE_CASE(18);
        else LANE_CASE(34);
    } else if (use_grid && self_sorted && b > 0 && v.qorder && nsample - 1 <= 64) {
        // large k, self queries: workgroup per cell over the LDS-staged neighbourhood; what it cannot decide goes through the
        // retry list to the general selection kernel
        int* retry_count = v.tie_count + 1;
        knn_cell_kernel<<<dim3(1024, b), 256, 0, stream>>>(nsample, xyz, offset, v.grids, v.cell_start, v.sorted, o, retry_count, v.qorder);
        knn_gridsel_kernel<<<1024, 256, 0, stream>>>(m, nsample, xyz, new_xyz, offset, new_offset, v.grids, v.cell_start, v.sorted, o, v.qorder,
                                                      retry_count);
    } else if (use_grid && nsample + 1 <= 128) {
        knn_gridsel_kernel<<<min(blocks, 65536), 256, 0, stream>>>(m, nsample, xyz, new_xyz, offset, new_offset, v.grids, v.cell_start, v.sorted, o,
                                                                    nullptr, nullptr);
    } else if (lane_ok && m >= 4 * lane_min) {
        if (nsample + 1 <= 66) LANE_CASE(66); else LANE_CASE(101);
    } else
#undef LANE_CASE
#undef PREF_CASE
    if (!use_grid && nsample + 1 <= 34 && (!ppf || group_idx) && m >= lane_brute_min) {
        // staging chunk: the mean cloud rounded up to 64 points, 64 .. 1024 (a larger cloud takes several chunks)
        int ch = b > 0 ? (int)(((long)n / b + 63) / 64 * 64) : 1024;
        ch = ch < 64 ? 64 : (ch > 1024 ? 1024 : ch);
#define LB_CASE(LC) knn_lane_brute_kernel<LC><<<div_up(m, 64), 64, (size_t)ch * 16, stream>>>(m, nsample, b, xyz, new_xyz, offset, new_offset, o, ch)
        const int need = nsample + 1;
        if (need <= 2) LB_CASE(2); else if (need <= 4) LB_CASE(4); else if (need <= 10) LB_CASE(10);
        else if (need <= 18) LB_CASE(18); else LB_CASE(34);
#undef LB_CASE
    } else if (use_grid) {
        if (nsample + 1 <= 64)
            knn_grid_kernel<1><<<blocks, 256, 0, stream>>>(m, nsample, xyz, new_xyz, offset, new_offset, v.grids, v.cell_start, v.sorted, o);
        else
            knn_grid_kernel<2><<<blocks, 256, 0, stream>>>(m, nsample, xyz, new_xyz, offset, new_offset, v.grids, v.cell_start, v.sorted, o);
    } else {
        if (nsample + 1 <= 64)
            knn_brute_kernel<1><<<blocks, 256, 0, stream>>>(m, nsample, xyz, new_xyz, offset, new_offset, o);
        else
            knn_brute_kernel<2><<<blocks, 256, 0, stream>>>(m, nsample, xyz, new_xyz, offset, new_offset, o);
    }
    roitr_prof_end(ROITR_PROF_KNN, stream);
    ROITR_LAUNCH_CHECK();
    roitr_prof_begin(ROITR_PROF_REPLAY, 0.0, stream);
    knn_replay_kernel<<<min(m, 1024), 64, 0, stream>>>(nsample, xyz, new_xyz, offset, new_offset, o);
    roitr_prof_end(ROITR_PROF_REPLAY, stream);
    ROITR_LAUNCH_CHECK();
    {   // debug (ROITR_KNN_STATS=1, synchronous): queries that went to the exact replay / from the cell kernel to the general one
        struct Stats {
            bool on = getenv("ROITR_KNN_STATS") != nullptr; long q = 0, ties = 0, retries = 0, calls = 0;
            ~Stats() { if (on) fprintf(stderr, "KNNSTATS calls %ld queries %ld replayed %ld cell->general %ld\n", calls, q, ties, retries); }
        };
        static Stats st;
        if (st.on) {
            int h[2] = {0, 0};
            ROITR_HIP(hipMemcpyAsync(h, v.tie_count, sizeof(h), hipMemcpyDeviceToHost, stream));
            ROITR_HIP(hipStreamSynchronize(stream));
            st.calls++; st.q += m; st.ties += h[0]; st.retries += h[1];
        }
    }
    return ROITR_OK;
}
}  // namespace

extern "C" int roitr_knnquery_ex(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz, const int* offset,
                                 const int* new_offset, int* idx, float* dist2, int* group_idx, float* ppf,
                                 const float* ref_normals, const float* query_normals, int use_grid, int m_capacity, void* ws,
                                 hipStream_t stream)
{
    return knnquery_impl(b, n, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, group_idx, ppf, ref_normals, query_normals, use_grid,
                         m_capacity, ws, INFINITY, stream);
}

// Exact drop-in for knnquery_cuda_kernel.h:9-17: no batch count, no point count, no workspace, void
// return, legacy default stream.  Brute-force path (segments are discovered on the device exactly as
// the reference does); the tie scratch is a process-global buffer grown on demand.
extern "C" void knnquery_cuda_launcher(int m, int nsample, const float* xyz, const float* new_xyz, const int* offset,
                                       const int* new_offset, int* idx, float* dist2)
{
    if (m <= 0) return;
    const size_t need = (size_t)m + 16;   // counters (4 ints) + tie list + alignment slack
    if (need > g_legacy_ws_ints) {
        if (g_legacy_ws) (void)hipFree(g_legacy_ws);
        g_legacy_ws_ints = need * 2;
        if (hipMalloc(&g_legacy_ws, g_legacy_ws_ints * sizeof(int)) != hipSuccess) {
            roitr_set_error("hipMalloc failed (legacy knn scratch)", __FILE__, __LINE__);
            g_legacy_ws = nullptr; g_legacy_ws_ints = 0;
            return;
        }
    }
    (void)roitr_knnquery_ex(0, 0, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, nullptr, nullptr, nullptr, nullptr, 0, m,
                            g_legacy_ws, nullptr);
}


// Nearest-neighbour distances for a radius test: dist2[q] is exact whenever it is < cap2; queries whose nearest
// reference point is farther than sqrt(cap2) get SOME value >= cap2 (the ring search stops as soon as everything unseen
// is beyond the cap).  lib/utils.py:509-521 only evaluates `nearest distance < overlap radius`.
extern "C" int roitr_knn_within(int b, int n, int m, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset,
                                float cap2, float* dist2, int use_grid, int m_capacity, void* ws, hipStream_t stream)
{
    return knnquery_impl(b, n, m, 1, xyz, new_xyz, offset, new_offset, nullptr, dist2, nullptr, nullptr, nullptr, nullptr, use_grid,
                         m_capacity, ws, cap2, stream);
}
