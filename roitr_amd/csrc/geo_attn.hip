// Global (superpoint-level) geometric transformer kernels.
//
// Reference: model/transformer/positional_encoding.py:94-154 (GeometricStructureEmbedding),
// model/transformer/geoattention.py:10-136 (MultiHeadAttention, RPEMultiHeadAttention).
//
// (1) geo_indices: per node i the distance indices d_ij/sigma_d and, for the 3 nearest neighbours
//     of i, the triplet angles a_ijk * 180/(sigma_a*pi).  pairwise_distance (l.9-34) is the matmul
//     form x2 - 2xy + y2 and its rounding noise is visible on the diagonal (sqrt of ~1e-7), so the
//     arithmetic is pinned to what torch-CPU produces for K = 3: xy is an fma chain in k order, the
//     squared norms are plain sums ((x0^2 + x1^2) + x2^2) -- verified against the golden vectors.
// (2) mha: one workgroup per query row, one wave per head, lane = key index.  The RPE branch is
//     folded:  q_h . (Wp_h e_ij + bp_h) = (Wp_h^T q_h) . e_ij + q_h . bp_h, and
//     sum_j a_ij (Wvp_h e_ij + bvp_h) = Wvp_h (sum_j a_ij e_ij) + bvp_h, so the (n,n,C) tensors
//     proj_p(E) / proj_vp(E) of geoattention.py:104-105 (n^2 C^2 MACs each, 3.3 GFLOP per self layer at
//     n = 78) are never formed; what remains on E is two streaming passes (n^2 C MACs each).
#include "common.h"
#include <cstdlib>
#include "prof.h"
#include "roitr_engine.h"

namespace {

// NB: hipcc's __fmul_rn/__fadd_rn are plain operators and would be re-fused by -ffp-contract=fast, so the
// functions that pin torch-CPU arithmetic switch contraction off explicitly.
__device__ __forceinline__ float sq_norm3(float x, float y, float z)
{
#pragma clang fp contract(off)
    const float a = x * x, b = y * y, c = z * z;
    return (a + b) + c;
}
__device__ __forceinline__ float dot3_fma(float ax, float ay, float az, float bx, float by, float bz)
{
#pragma clang fp contract(off)
    const float m = ax * bx;
    return __fmaf_rn(az, bz, __fmaf_rn(ay, by, m));
}
// pairwise_distance(x, y)[i][j] clamped at 0 (positional_encoding.py:30-33)
__device__ __forceinline__ float pair_sqdist(float ax, float ay, float az, float bx, float by, float bz)
{
#pragma clang fp contract(off)
    const float x2 = sq_norm3(ax, ay, az), y2 = sq_norm3(bx, by, bz);
    const float xy = dot3_fma(ax, ay, az, bx, by, bz);
    const float t = 2.0f * xy;
    const float u = x2 - t;
    return fmaxf(u + y2, 0.0f);
}

// grid: one block per node row (global row index over all clouds); 256 threads over j
__global__ __launch_bounds__(256) void geo_indices_kernel(const float* __restrict__ pts, const int* __restrict__ offset,
                                                          const int* __restrict__ cloud_of_row, const long* __restrict__ eoff,
                                                          float inv_sigma_d_is_div, float sigma_d, float factor_a, int angle_k,
                                                          float* __restrict__ d_idx, float* __restrict__ a_idx)
{
    __shared__ float sd[1024];
    __shared__ unsigned long long red[4];
    __shared__ int nb[8];
    const int row = blockIdx.x;
    const int c = cloud_of_row[row];
    const int s = c == 0 ? 0 : offset[c - 1], e = offset[c];
    const int n = e - s, i = row - s;
    const float px = pts[(size_t)row * 3], py = pts[(size_t)row * 3 + 1], pz = pts[(size_t)row * 3 + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* drow = d_idx + eoff[c] + (size_t)i * n;
    for (int j = tid; j < n; j += 256) {
        const float* q = pts + (size_t)(s + j) * 3;
        const float d = sqrtf(pair_sqdist(px, py, pz, q[0], q[1], q[2]));
        sd[j] = d;
        drow[j] = d / sigma_d;
    }
    __syncthreads();
    // k+1 smallest of the row, ascending, lowest index first on ties (topk(largest=False), l.124); drop the first
    for (int t = 0; t <= angle_k; ++t) {
        unsigned long long best = ~0ull;
        for (int j = tid; j < n; j += 256) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(sd[j]) << 32) | (unsigned)j;
            best = key < best ? key : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long w = __shfl_xor(best, o, 64); best = w < best ? w : best; }
        if (lane == 0) red[wave] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long b = red[0];
            for (int w = 1; w < 4; ++w) b = red[w] < b ? red[w] : b;
            const int j = (int)(unsigned)b;
            nb[t] = (b == ~0ull) ? i : j;
            if (b != ~0ull) sd[j] = INFINITY;  // distances are >= 0: +inf bits sort last
        }
        __syncthreads();
    }
    (void)inv_sigma_d_is_div;
    float* arow = a_idx + (eoff[c] + (size_t)i * n) * angle_k;
    for (int j = tid; j < n; j += 256) {
        const float* q = pts + (size_t)(s + j) * 3;
        const float ax = q[0] - px, ay = q[1] - py, az = q[2] - pz;  // anc = p_j - p_i (l.129)
        for (int k = 0; k < angle_k; ++k) {
            const float* r = pts + (size_t)(s + nb[k + 1]) * 3;
            const float rx = r[0] - px, ry = r[1] - py, rz = r[2] - pz;  // ref = knn_k - p_i (l.128)
            const float cx = ry * az - rz * ay, cy = rz * ax - rx * az, cz = rx * ay - ry * ax;
            const float sn = sqrtf(cx * cx + cy * cy + cz * cz);
            const float cs = 0.0f + rx * ax + ry * ay + rz * az;  // torch.sum starts from +0: (+0) + (-0) = +0
            arow[(size_t)j * angle_k + k] = atan2f(sn, cs) * factor_a;
        }
    }
}

// ------------------------------------------------------------------ multi-head attention, one block per query row
__global__ __launch_bounds__(256) void mha_kernel(RoitrMha a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware row order: every XCD (private L2) gets a contiguous range of query rows, so the key / value rows of a cloud
    // are fetched into ONE L2 instead of all eight (PMC before: 1.27x the algorithmic bytes on the self layers)
    const int rowi = xcd_block_id(a.q_rows);
    if (rowi >= a.q_rows) return;
    const int row = a.q_row0 + rowi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C, NH = a.heads, c = C / NH;
    const int cl = a.cloud_of_row[row];
    const int kc = a.partner ? a.partner[cl] : cl;
    const int ks = kc == 0 ? 0 : a.offset[kc - 1], ke = a.offset[kc];
    const int nk = ke - ks;
    const int qs_ = cl == 0 ? 0 : a.offset[cl - 1];
    const int qi = row - qs_;  // index inside its own cloud (self attention: the diagonal position)

    float* qsh = smem;                      // C        : q row
    float* qt = qsh + C;                    // NH*C     : folded rpe queries (only with E)
    float* sc = qt + (a.E ? NH * C : 0);    // NH*nkmax : scores -> probabilities
    float* sc2 = sc + NH * a.nk_max;        // NH*nkmax : diagonal-masked probabilities (only with E)
    float* qb = sc2 + (a.E ? NH * a.nk_max : 0);  // 8 : q_h . bp_h

    const float* qrow = a.q + (size_t)row * a.ldq;
    for (int i = tid; i < C; i += 256) qsh[i] = qrow[i];
    if (a.E) {
        const float* qtr = a.qt + (size_t)row * NH * C;
        for (int i = tid; i < NH * C; i += 256) qt[i] = qtr[i];
    }
    __syncthreads();
    if (a.E && tid < NH) {
        float s = 0.f;
        for (int i = 0; i < c; ++i) s += qsh[tid * c + i] * a.bp[tid * c + i];
        qb[tid] = s;
    }
    __syncthreads();

    const float* Erow = a.E ? a.E + a.eoff[cl] * C + (size_t)qi * nk * C : nullptr;
    // ---- scores: wave = head (loop if NH > 4), lane = key
    for (int h = wave; h < NH; h += 4) {
        const float* qh = qsh + h * c;
        for (int j0 = 0; j0 < nk; j0 += 64) {
            const int j = j0 + lane;
            if (j < nk) {
                const float* krow = a.k + (size_t)(ks + j) * a.ldk + h * c;
                float dot = 0.f;
                for (int i = 0; i < c; i += 4) {
                    const float4 kv = *reinterpret_cast<const float4*>(krow + i);
                    const float4 qv = *reinterpret_cast<const float4*>(qh + i);
                    dot += kv.x * qv.x; dot += kv.y * qv.y; dot += kv.z * qv.z; dot += kv.w * qv.w;
                }
                if (Erow) {
                    const float* er = Erow + (size_t)j * C;
                    const float* qth = qt + h * C;
                    float dp = 0.f;
                    for (int i = 0; i < C; i += 4) {
                        const float4 ev = *reinterpret_cast<const float4*>(er + i);
                        const float4 qv = *reinterpret_cast<const float4*>(qth + i);
                        dp += ev.x * qv.x; dp += ev.y * qv.y; dp += ev.z * qv.z; dp += ev.w * qv.w;
                    }
                    dot += dp + qb[h];
                }
                sc[h * a.nk_max + j] = dot * a.scale;
            }
        }
        // softmax (and the diagonal-masked softmax of geoattention.py:117-134) over j, inside this wave
        float mx = -INFINITY, mx2 = -INFINITY;
        for (int j = lane; j < nk; j += 64) {
            const float v = sc[h * a.nk_max + j];
            mx = fmaxf(mx, v);
            if (j != qi) mx2 = fmaxf(mx2, v);
        }
        mx = wave_max(mx); mx2 = wave_max(mx2);
        float sm = 0.f, sm2 = 0.f;
        for (int j = lane; j < nk; j += 64) {
            const float v = sc[h * a.nk_max + j];
            const float e1 = expf(v - mx);
            sc[h * a.nk_max + j] = e1; sm += e1;
            if (a.E) {
                const float e2 = j != qi ? expf(v - mx2) : 0.f;
                sc2[h * a.nk_max + j] = e2; sm2 += e2;
            }
        }
        sm = wave_sum(sm); sm2 = wave_sum(sm2);
        for (int j = lane; j < nk; j += 64) {
            sc[h * a.nk_max + j] /= sm;
            if (a.E) sc2[h * a.nk_max + j] /= sm2;
        }
    }
    __syncthreads();
    // ---- hidden[h*c + ch] = sum_j p[h][j] v[j][h*c + ch]
    for (int ch = tid; ch < C; ch += 256) {
        const int h = ch / c;
        float acc = 0.f;
        for (int j = 0; j < nk; ++j) acc += sc[h * a.nk_max + j] * a.v[(size_t)(ks + j) * a.ldv + ch];
        a.out[(size_t)row * a.ldo + ch] = acc;
    }
    // ---- ebar[h][:] = sum_j p2[h][j] E[i][j][:]
    if (Erow) {
        for (int ch = tid; ch < C; ch += 256) {
            float acc[8];
#pragma unroll
            for (int h = 0; h < 8; ++h) acc[h] = 0.f;
            for (int j = 0; j < nk; ++j) {
                const float ev = Erow[(size_t)j * C + ch];
#pragma unroll
                for (int h = 0; h < 8; ++h)
                    if (h < NH) acc[h] += sc2[h * a.nk_max + j] * ev;
            }
#pragma unroll
            for (int h = 0; h < 8; ++h)
                if (h < NH) a.ebar[((size_t)row * NH + h) * C + ch] = acc[h];
        }
    }
}


// ------------------------------------------------------------------ self attention with the folded RPE branch, C = 256, 4 heads
// One block per query row i, same math as mha_kernel.  The (n, C) slab E[i, :, :] -- the only HBM-sized operand, n*C*4
// = 80 KB at n = 78 -- is read from memory exactly ONCE, fully coalesced (wave w owns keys j = w, w+4, ...; a lane holds
// the float4 of channels 4*lane.. of each of its rows), and stays in registers for both uses: the score pass
// (q~_h . E_ij, wave reductions on the DPP path) and the value pass (ebar_h = sum_j a'_hj E_ij, register FMAs; the four
// waves' partial sums meet in LDS).  Channels 4*lane..4*lane+3 belong to head lane/16 = the lane's DPP row, so q_h . k_j
// is a row reduction.  mha_kernel streams E twice with one 1 KB row per LANE (uncoalesced, L1-thrashing): 1.9 ms per
// launch at 128 pairs against 0.3-0.4 ms for one pass over E at HBM speed.
__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x))); }

template <int R>   // R = key rows per wave kept in registers: n <= 4 R
__global__ __launch_bounds__(256, R <= 20 ? 3 : 2) void mha_geo_kernel(RoitrMha a)
{
    constexpr int C = 256, NH = 4, NKP = 4 * R;
    __shared__ __attribute__((aligned(16))) float sc[NH][NKP];    // scores -> probabilities [head][key]
    __shared__ __attribute__((aligned(16))) float sc2t[NKP * 4];   // diagonal-masked probabilities [key][head]
    __shared__ __attribute__((aligned(16))) float red[4][NH * C];  // per-wave ebar partials
    __shared__ __attribute__((aligned(16))) float redh[4][C];      // per-wave hidden partials
    // XCD-aware row order: every XCD (private L2) gets a contiguous range of query rows, so the key / value rows of a cloud
    // are fetched into ONE L2 instead of all eight (PMC before: 1.27x the algorithmic bytes on the self layers)
    const int rowi = xcd_block_id(a.q_rows);
    if (rowi >= a.q_rows) return;
    const int row = a.q_row0 + rowi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 4;
    const int cl = a.cloud_of_row[row];
    const int ks = cl == 0 ? 0 : a.offset[cl - 1], nk = a.offset[cl] - ks;
    const int qi = row - ks;
    const float* Erow = a.E + a.eoff[cl] * C + (size_t)qi * nk * C;
    float4 e[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int j = wave + 4 * rr;
        e[rr] = j < nk ? reinterpret_cast<const float4*>(Erow + (size_t)j * C)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 qv = reinterpret_cast<const float4*>(a.q + (size_t)row * a.ldq)[lane];
    float4 qt4[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) qt4[h] = reinterpret_cast<const float4*>(a.qt + ((size_t)row * NH + h) * C)[lane];
    // this lane's share of q_h . bp_h (h = its DPP row): summed over the row inside the score reduction below
    const float qb_l = dot4(qv, reinterpret_cast<const float4*>(a.bp)[lane]);
    // ---- scores, part 1 (while the E rows are in flight): this lane's share of q_h . k_j + q_h . bp_h from L2-resident key rows,
    // 5 rows per batch
    float qk[R];
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += 5) {
        float4 kv[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int j = wave + 4 * (r0 + u);
            kv[u] = reinterpret_cast<const float4*>(a.k + (size_t)(ks + (j < nk ? j : nk - 1)) * a.ldk)[lane];
        }
#pragma unroll
        for (int u = 0; u < 5; ++u)
            if (r0 + u < R) qk[r0 + u] = dot4(qv, kv[u]) + qb_l;
    }
    // ---- scores, part 2: + q~_h . E_ij.  The 4 heads x 16 key rows = 64 wave-wide dot products of a batch of rows are reduced
    // TOGETHER: v_permlane32_swap / v_permlane16_swap fold the four DPP rows while row h keeps head h's quarter of the values,
    // row16_transpose_sum finishes inside the row -- lane (h, i) ends with the score of (head h, key row16_slot(i)) in 63 adds
    // (round 2: one 64-lane reduction per value; first form of round 3: groups of 16 values + two LDS shuffles per group).  The
    // terms that do not involve E ride in the same sum: a lane adds its share of q_h . k_j + q_h . bp_h to the value of its own head.
    static_assert(R % 4 == 0, "key rows per wave come in groups of four");
    {
        const int i16 = lane & 15;
#pragma unroll
        for (int b16 = 0; b16 + 16 <= R; b16 += 16) {
            float z[16];
            {
                float w[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int h0 = i >> 4, r = i & 15;      // value i = (head h0, row r), value i + 32 = (head h0 + 2, row r)
                    const float4 ev = e[b16 + r];
                    float d0 = hl == h0 ? qk[b16 + r] : 0.f, d1 = hl == h0 + 2 ? qk[b16 + r] : 0.f;
                    d0 = fmaf(qt4[h0].x, ev.x, d0); d0 = fmaf(qt4[h0].y, ev.y, d0); d0 = fmaf(qt4[h0].z, ev.z, d0); d0 = fmaf(qt4[h0].w, ev.w, d0);
                    d1 = fmaf(qt4[h0 + 2].x, ev.x, d1); d1 = fmaf(qt4[h0 + 2].y, ev.y, d1); d1 = fmaf(qt4[h0 + 2].z, ev.z, d1); d1 = fmaf(qt4[h0 + 2].w, ev.w, d1);
                    w[i] = swap32_sum(d0, d1);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = swap16_sum(w[i], w[i + 16]);
            }
            const float tot = row16_transpose_sum(z, lane);
            const int j = wave + 4 * (b16 + row16_slot(i16));
            if (j < nk) sc[hl][j] = tot * a.scale;
        }
        // the rows past the last full batch (R = 20: four of them): groups of 4 heads x 4 rows, row sums + two cross-row adds
#pragma unroll
        for (int g4 = (R / 16) * 4; g4 < R / 4; ++g4) {
            float v[16];
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 ev = e[4 * g4 + r];
                    float d = hl == h ? qk[4 * g4 + r] : 0.f;
                    d = fmaf(qt4[h].x, ev.x, d); d = fmaf(qt4[h].y, ev.y, d); d = fmaf(qt4[h].z, ev.z, d); d = fmaf(qt4[h].w, ev.w, d);
                    v[h * 4 + r] = d;
                }
            float tot = row16_transpose_sum(v, lane);
            tot += __shfl_xor(tot, 16, 64);
            tot += __shfl_xor(tot, 32, 64);
            if (lane < 16) {
                const int slot = row16_slot(i16), h = slot >> 2, j = wave + 4 * (4 * g4 + (slot & 3));
                if (j < nk) sc[h][j] = tot * a.scale;
            }
        }
    }
    // the first EB of this wave's value rows: in flight across the softmax (more would spill: the E rows stay in registers)
    constexpr int EB = 4;
    float4 vfirst[EB];
#pragma unroll
    for (int rr = 0; rr < EB; ++rr) {
        const int j = wave + 4 * rr;
        vfirst[rr] = reinterpret_cast<const float4*>(a.v + (size_t)(ks + (j < nk ? j : nk - 1)) * a.ldv)[lane];
    }
    __syncthreads();
    // ---- softmax and the diagonal-masked softmax (geoattention.py:117-134) over the keys: wave = head
    {
        const int h = wave;
        float mx = -INFINITY, mx2 = -INFINITY;
        for (int j = lane; j < nk; j += 64) {
            const float v = sc[h][j];
            mx = fmaxf(mx, v);
            if (j != qi) mx2 = fmaxf(mx2, v);
        }
        mx = wave_max(mx); mx2 = wave_max(mx2);
        float sm = 0.f, sm2 = 0.f;
        float e1[(NKP + 63) / 64], e2[(NKP + 63) / 64];
#pragma unroll
        for (int u = 0; u < (NKP + 63) / 64; ++u) {
            const int j = lane + 64 * u;
            const float v = j < nk ? sc[h][j] : 0.f;
            e1[u] = j < nk ? expf(v - mx) : 0.f;
            e2[u] = (j < nk && j != qi) ? expf(v - mx2) : 0.f;
            sm += e1[u]; sm2 += e2[u];
        }
        sm = wave_sum(sm); sm2 = wave_sum(sm2);
        // rows past the cloud's last key get probability 0 in both tables: the passes below need no per-key bound checks
#pragma unroll
        for (int u = 0; u < (NKP + 63) / 64; ++u) {
            const int j = lane + 64 * u;
            if (j < NKP) { sc[h][j] = j < nk ? e1[u] / sm : 0.f; sc2t[j * 4 + h] = j < nk ? e2[u] / sm2 : 0.f; }
        }
    }
    __syncthreads();
    // ---- hidden[ch] = sum_j p[head(ch)][j] v[j][ch].  Round 3: same key split as E (wave w owns keys w, w+4, ...; a lane the
    // float4 of channels 4*lane.., whose head is its DPP row) -- the value rows arrive as R coalesced 1 KB loads per wave in
    // a few batches, the first one requested BEFORE the softmax (above), instead of n/8 dependent batches of 8 strided scalar
    // loads per thread (10 serial L2 round trips per query row: half of the kernel's time at n = 78); partial sums of the four
    // waves meet in LDS like ebar's.
    float4 hacc = make_float4(0.f, 0.f, 0.f, 0.f);
    {
#pragma unroll
        for (int rr = 0; rr < EB; ++rr) {
            const int j = wave + 4 * rr;
            const float p = sc[hl][j];
            hacc.x = fmaf(p, vfirst[rr].x, hacc.x); hacc.y = fmaf(p, vfirst[rr].y, hacc.y); hacc.z = fmaf(p, vfirst[rr].z, hacc.z); hacc.w = fmaf(p, vfirst[rr].w, hacc.w);
        }
        constexpr int CH = 8;   // rows per later batch (a real loop: unrolled, the compiler hoists every batch's loads and spills)
#pragma unroll 1
        for (int r0 = EB; r0 < R; r0 += CH) {
            float4 vb[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int j = wave + 4 * (r0 + u);
                vb[u] = reinterpret_cast<const float4*>(a.v + (size_t)(ks + (j < nk ? j : nk - 1)) * a.ldv)[lane];
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int j = wave + 4 * (r0 + u);
                const float p = r0 + u < R ? sc[hl][j < NKP ? j : 0] : 0.f;
                hacc.x = fmaf(p, vb[u].x, hacc.x); hacc.y = fmaf(p, vb[u].y, hacc.y); hacc.z = fmaf(p, vb[u].z, hacc.z); hacc.w = fmaf(p, vb[u].w, hacc.w);
            }
        }
        reinterpret_cast<float4*>(redh[wave])[lane] = hacc;
    }
    // ---- ebar[h][:] = sum_j p2[h][j] E[i][j][:]: this wave's rows out of registers, then the 4 waves through LDS
    float4 acc[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) acc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int j = wave + 4 * rr;
        const float4 p2 = reinterpret_cast<const float4*>(sc2t)[j];
        const float w0 = p2.x, w1 = p2.y, w2 = p2.z, w3 = p2.w;
        acc[0].x = fmaf(w0, e[rr].x, acc[0].x); acc[0].y = fmaf(w0, e[rr].y, acc[0].y); acc[0].z = fmaf(w0, e[rr].z, acc[0].z); acc[0].w = fmaf(w0, e[rr].w, acc[0].w);
        acc[1].x = fmaf(w1, e[rr].x, acc[1].x); acc[1].y = fmaf(w1, e[rr].y, acc[1].y); acc[1].z = fmaf(w1, e[rr].z, acc[1].z); acc[1].w = fmaf(w1, e[rr].w, acc[1].w);
        acc[2].x = fmaf(w2, e[rr].x, acc[2].x); acc[2].y = fmaf(w2, e[rr].y, acc[2].y); acc[2].z = fmaf(w2, e[rr].z, acc[2].z); acc[2].w = fmaf(w2, e[rr].w, acc[2].w);
        acc[3].x = fmaf(w3, e[rr].x, acc[3].x); acc[3].y = fmaf(w3, e[rr].y, acc[3].y); acc[3].z = fmaf(w3, e[rr].z, acc[3].z); acc[3].w = fmaf(w3, e[rr].w, acc[3].w);
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) reinterpret_cast<float4*>(red[wave])[h * 64 + lane] = acc[h];
    __syncthreads();
    a.out[(size_t)row * a.ldo + tid] = (redh[0][tid] + redh[1][tid]) + (redh[2][tid] + redh[3][tid]);
#pragma unroll
    for (int h = 0; h < NH; ++h)
        a.ebar[((size_t)row * NH + h) * C + tid] = (red[0][h * C + tid] + red[1][h * C + tid]) + (red[2][h * C + tid] + red[3][h * C + tid]);
}


// ------------------------------------------------------------------ the same for clouds with more than 128 superpoints
// (up to the engine's 1024: 30000-point clouds have 468).  The E slab of a query row no longer fits the registers of a
// block (n x 1 KB), so it is streamed twice -- still one coalesced float4 per lane and row, four rows in flight per
// wave, the second pass mostly out of L2 / Infinity Cache -- instead of the generic kernel's one-row-per-lane walk.
__global__ __launch_bounds__(256) void mha_geo_stream_kernel(RoitrMha a)
{
    constexpr int C = 256, NH = 4, NKP = 1024;
    __shared__ __attribute__((aligned(16))) float sc[NH][NKP];
    __shared__ __attribute__((aligned(16))) float sc2t[NKP * 4];
    __shared__ __attribute__((aligned(16))) float red[4][NH * C];
    // XCD-aware row order: every XCD (private L2) gets a contiguous range of query rows, so the key / value rows of a cloud
    // are fetched into ONE L2 instead of all eight (PMC before: 1.27x the algorithmic bytes on the self layers)
    const int rowi = xcd_block_id(a.q_rows);
    if (rowi >= a.q_rows) return;
    const int row = a.q_row0 + rowi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 4;
    const int cl = a.cloud_of_row[row];
    const int ks = cl == 0 ? 0 : a.offset[cl - 1], nk = a.offset[cl] - ks;
    const int qi = row - ks;
    const float* Erow = a.E + a.eoff[cl] * C + (size_t)qi * nk * C;
    const float4 qv = reinterpret_cast<const float4*>(a.q + (size_t)row * a.ldq)[lane];
    float4 qt4[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) qt4[h] = reinterpret_cast<const float4*>(a.qt + ((size_t)row * NH + h) * C)[lane];
    const float qb = row_allsum(dot4(qv, reinterpret_cast<const float4*>(a.bp)[lane]));
    // ---- scores: 4 key rows of this wave per trip (E and k rows requested together)
    for (int j0 = wave; j0 < nk; j0 += 16) {
        float4 ev[4], kv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = min(j0 + 4 * u, nk - 1);
            ev[u] = reinterpret_cast<const float4*>(Erow + (size_t)j * C)[lane];
            kv[u] = reinterpret_cast<const float4*>(a.k + (size_t)(ks + j) * a.ldk)[lane];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 4 * u;
            const float s1 = row_allsum(dot4(qv, kv[u]));
            const float se0 = wave_sum(dot4(qt4[0], ev[u])), se1 = wave_sum(dot4(qt4[1], ev[u]));
            const float se2 = wave_sum(dot4(qt4[2], ev[u])), se3 = wave_sum(dot4(qt4[3], ev[u]));
            const float se = hl == 0 ? se0 : (hl == 1 ? se1 : (hl == 2 ? se2 : se3));
            if (j < nk && (lane & 15) == 0) sc[hl][j] = (s1 + (se + qb)) * a.scale;
        }
    }
    __syncthreads();
    {   // softmax and diagonal-masked softmax: wave = head
        const int h = wave;
        float mx = -INFINITY, mx2 = -INFINITY;
        for (int j = lane; j < nk; j += 64) { const float v = sc[h][j]; mx = fmaxf(mx, v); if (j != qi) mx2 = fmaxf(mx2, v); }
        mx = wave_max(mx); mx2 = wave_max(mx2);
        float sm = 0.f, sm2 = 0.f;
        for (int j = lane; j < nk; j += 64) {
            const float v = sc[h][j];
            const float e1 = expf(v - mx), e2 = j != qi ? expf(v - mx2) : 0.f;
            sc[h][j] = e1; sc2t[j * 4 + h] = e2; sm += e1; sm2 += e2;
        }
        sm = wave_sum(sm); sm2 = wave_sum(sm2);
        for (int j = lane; j < nk; j += 64) { sc[h][j] /= sm; sc2t[j * 4 + h] /= sm2; }
    }
    __syncthreads();
    {   // hidden: thread = channel
        const int h = tid >> 6;
        const float* vp = a.v + (size_t)ks * a.ldv + tid;
        float acc = 0.f;
        int j = 0;
        for (; j + 8 <= nk; j += 8) {
            float vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = vp[(size_t)(j + u) * a.ldv];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(sc[h][j + u], vv[u], acc);
        }
        for (; j < nk; ++j) acc = fmaf(sc[h][j], vp[(size_t)j * a.ldv], acc);
        a.out[(size_t)row * a.ldo + tid] = acc;
    }
    // ---- ebar: second pass over this wave's E rows
    float4 acc[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) acc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = wave; j0 < nk; j0 += 16) {
        float4 ev[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) ev[u] = reinterpret_cast<const float4*>(Erow + (size_t)min(j0 + 4 * u, nk - 1) * C)[lane];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 4 * u;
            const float4 p2 = reinterpret_cast<const float4*>(sc2t)[j < nk ? j : 0];
            const float w[4] = {j < nk ? p2.x : 0.f, j < nk ? p2.y : 0.f, j < nk ? p2.z : 0.f, j < nk ? p2.w : 0.f};
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                acc[h].x = fmaf(w[h], ev[u].x, acc[h].x); acc[h].y = fmaf(w[h], ev[u].y, acc[h].y);
                acc[h].z = fmaf(w[h], ev[u].z, acc[h].z); acc[h].w = fmaf(w[h], ev[u].w, acc[h].w);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) reinterpret_cast<float4*>(red[wave])[h * 64 + lane] = acc[h];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NH; ++h)
        a.ebar[((size_t)row * NH + h) * C + tid] = (red[0][h * C + tid] + red[1][h * C + tid]) + (red[2][h * C + tid] + red[3][h * C + tid]);
}

// ------------------------------------------------------------------ plain multi-head attention (cross layers), C = 256, 4 heads
// geoattention.py:26-66 on the same lane = channel-quad layout as mha_geo_kernel: key rows are read coalesced (one
// float4 per lane, head = DPP row), the four waves split the keys, values are accumulated by thread = channel.
// Round 3: a block takes QB = 4 consecutive query rows.  With one row per block every query re-read all key and value rows of the
// partner cloud from L2 (78 x 2 KB per query: 6.2 GB per launch at 512 pairs = 28 TB/s -- the kernel ran at the L2's bandwidth, not
// at HBM's); rows of the same cloud now share each loaded row (4 dot products / 4 accumulations per load).  A block whose rows
// straddle a cloud boundary works through its runs of equal cloud one after the other.  Per query the arithmetic and its order are
// those of the one-row form: results are bit-identical.
template <int NKP>   // keys per cloud bound (LDS score rows, softmax registers)
__global__ __launch_bounds__(256) void mha_plain_kernel(RoitrMha a)
{
    constexpr int NH = 4, QB = 4;
    __shared__ float sc[QB][NH][NKP];
    // XCD-aware row order: every XCD (private L2) gets a contiguous range of query rows, so the key / value rows of a cloud
    // are fetched into ONE L2 instead of all eight (PMC before: 1.27x the algorithmic bytes on the self layers)
    const int nblk = (a.q_rows + QB - 1) / QB;
    const int blk = xcd_block_id(nblk);
    if (blk >= nblk) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 4;
    const int r_first = blk * QB, r_end = min(r_first + QB, a.q_rows);
    for (int r0 = r_first; r0 < r_end;) {
        const int row0 = a.q_row0 + r0;
        const int cl = a.cloud_of_row[row0];
        int nq = 1;                                            // rows of this run: same cloud as row0 (block-uniform)
        while (r0 + nq < r_end && a.cloud_of_row[row0 + nq] == cl) ++nq;
        const int kc = a.partner ? a.partner[cl] : cl;
        const int ks = kc == 0 ? 0 : a.offset[kc - 1], nk = a.offset[kc] - ks;
        float4 qv[QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) qv[q] = reinterpret_cast<const float4*>(a.q + (size_t)(row0 + (q < nq ? q : 0)) * a.ldq)[lane];
        const float* kbase = a.k + (size_t)ks * a.ldk + lane * 4;
        for (int j0 = wave; j0 < nk; j0 += 16) {   // 4 keys of this wave per trip, loads issued together
            float4 kv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 4 * u;
                kv[u] = *reinterpret_cast<const float4*>(kbase + (size_t)(j < nk ? j : nk - 1) * a.ldk);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 4 * u;
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    const float s = row_allsum(dot4(qv[q], kv[u]));
                    if (j < nk && (lane & 15) == 0) sc[q][hl][j] = s * a.scale;
                }
            }
        }
        __syncthreads();
        {   // softmax over the keys: wave = head, one query after the other
            const int h = wave;
            for (int q = 0; q < nq; ++q) {
                float e1[NKP / 64];
                float mx = -INFINITY;
#pragma unroll
                for (int u = 0; u < NKP / 64; ++u) { const int j = lane + 64 * u; e1[u] = j < nk ? sc[q][h][j] : -INFINITY; mx = fmaxf(mx, e1[u]); }
                mx = wave_max(mx);
                float sm = 0.f;
#pragma unroll
                for (int u = 0; u < NKP / 64; ++u) { const int j = lane + 64 * u; e1[u] = j < nk ? expf(e1[u] - mx) : 0.f; sm += e1[u]; }
                sm = wave_sum(sm);
#pragma unroll
                for (int u = 0; u < NKP / 64; ++u) { const int j = lane + 64 * u; if (j < nk) sc[q][h][j] = e1[u] / sm; }
            }
        }
        __syncthreads();
        {
            const int h = tid >> 6;
            const float* vp = a.v + (size_t)ks * a.ldv + tid;
            float acc[QB];
#pragma unroll
            for (int q = 0; q < QB; ++q) acc[q] = 0.f;
            int j = 0;
            for (; j + 8 <= nk; j += 8) {
                float vv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) vv[u] = vp[(size_t)(j + u) * a.ldv];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int q = 0; q < QB; ++q) acc[q] = fmaf(sc[q][h][j + u], vv[u], acc[q]);
            }
            for (; j < nk; ++j) {
                const float vv = vp[(size_t)j * a.ldv];
#pragma unroll
                for (int q = 0; q < QB; ++q) acc[q] = fmaf(sc[q][h][j], vv, acc[q]);
            }
#pragma unroll
            for (int q = 0; q < QB; ++q)
                if (q < nq) a.out[(size_t)(row0 + q) * a.ldo + tid] = acc[q];
        }
        r0 += nq;
        if (r0 < r_end) __syncthreads();   // the score table is re-used by the next run
    }
}

// ------------------------------------------------------------------ the same two kernels for C = 256 CQ (CQ = 2: the factor-2
// width of the 4DMatch configuration, model/RIGA_v2.py:24-28) and / or an E tensor stored in bf16 (engine operand_dtype =
// bf16).  Same layout idea: a lane owns 4 CQ consecutive channels (all inside head lane / 16), rows are read coalesced,
// E is streamed twice (the slab of a query row is n x 2 KB at C = 512: it does not fit the registers of a block).
// Before these existed the factor-2 configuration ran on the generic mha_kernel (one row per lane, uncoalesced).
template <int CQ> __device__ __forceinline__ void ldrow(const float* base, int lane, float4 (&d)[CQ])
{
#pragma unroll
    for (int c = 0; c < CQ; ++c) d[c] = reinterpret_cast<const float4*>(base)[lane * CQ + c];
}
// bf16 row: 4 CQ consecutive bf16 per lane, widened exactly
template <int CQ> __device__ __forceinline__ void ldrow_h(const unsigned short* base, int lane, float4 (&d)[CQ])
{
#pragma unroll
    for (int c = 0; c < CQ; ++c) {
        const uint2 u = reinterpret_cast<const uint2*>(base)[lane * CQ + c];
        d[c] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
}
template <int CQ> __device__ __forceinline__ float dotq(const float4 (&x)[CQ], const float4 (&y)[CQ])
{
    float s = dot4(x[0], y[0]);
#pragma unroll
    for (int c = 1; c < CQ; ++c) s += dot4(x[c], y[c]);
    return s;
}

template <int CQ, bool EH>
__global__ __launch_bounds__(256) void mha_geo_wide_kernel(RoitrMha a)
{
    constexpr int C = 256 * CQ, NH = 4, NKP = 512;
    __shared__ __attribute__((aligned(16))) float sc[NH][NKP];
    __shared__ __attribute__((aligned(16))) float sc2t[NKP * 4];
    __shared__ __attribute__((aligned(16))) float red[4][NH * C];
    const int rowi = xcd_block_id(a.q_rows);
    if (rowi >= a.q_rows) return;
    const int row = a.q_row0 + rowi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 4;
    const int cl = a.cloud_of_row[row];
    const int ks = cl == 0 ? 0 : a.offset[cl - 1], nk = a.offset[cl] - ks;
    const int qi = row - ks;
    const size_t e0 = (size_t)a.eoff[cl] * C + (size_t)qi * nk * C;   // element offset of E[i, 0, 0]
    const float* Ef = a.E + e0;
    const unsigned short* Eh = reinterpret_cast<const unsigned short*>(a.E) + e0;
    auto lde = [&](int j, float4 (&d)[CQ]) {
        if (EH) ldrow_h<CQ>(Eh + (size_t)j * C, lane, d);
        else ldrow<CQ>(Ef + (size_t)j * C, lane, d);
    };
    float4 qv[CQ], qt4[NH][CQ], bp4[CQ];
    ldrow<CQ>(a.q + (size_t)row * a.ldq, lane, qv);
#pragma unroll
    for (int h = 0; h < NH; ++h) ldrow<CQ>(a.qt + ((size_t)row * NH + h) * C, lane, qt4[h]);
    ldrow<CQ>(a.bp, lane, bp4);
    const float qb = row_allsum(dotq<CQ>(qv, bp4));   // q_h . bp_h, h = this lane's row
    // ---- scores: 4 key rows of this wave per trip (E and k rows requested together)
    for (int j0 = wave; j0 < nk; j0 += 16) {
        float4 ev[4][CQ], kv[4][CQ];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = min(j0 + 4 * u, nk - 1);
            lde(j, ev[u]);
            ldrow<CQ>(a.k + (size_t)(ks + j) * a.ldk, lane, kv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 4 * u;
            const float s1 = row_allsum(dotq<CQ>(qv, kv[u]));
            const float se0 = wave_sum(dotq<CQ>(qt4[0], ev[u])), se1 = wave_sum(dotq<CQ>(qt4[1], ev[u]));
            const float se2 = wave_sum(dotq<CQ>(qt4[2], ev[u])), se3 = wave_sum(dotq<CQ>(qt4[3], ev[u]));
            const float se = hl == 0 ? se0 : (hl == 1 ? se1 : (hl == 2 ? se2 : se3));
            if (j < nk && (lane & 15) == 0) sc[hl][j] = (s1 + (se + qb)) * a.scale;
        }
    }
    __syncthreads();
    {   // softmax and diagonal-masked softmax (geoattention.py:117-134): wave = head
        const int h = wave;
        float mx = -INFINITY, mx2 = -INFINITY;
        for (int j = lane; j < nk; j += 64) { const float v = sc[h][j]; mx = fmaxf(mx, v); if (j != qi) mx2 = fmaxf(mx2, v); }
        mx = wave_max(mx); mx2 = wave_max(mx2);
        float sm = 0.f, sm2 = 0.f;
        for (int j = lane; j < nk; j += 64) {
            const float v = sc[h][j];
            const float e1 = expf(v - mx), e2 = j != qi ? expf(v - mx2) : 0.f;
            sc[h][j] = e1; sc2t[j * 4 + h] = e2; sm += e1; sm2 += e2;
        }
        sm = wave_sum(sm); sm2 = wave_sum(sm2);
        for (int j = lane; j < nk; j += 64) { sc[h][j] /= sm; sc2t[j * 4 + h] /= sm2; }
    }
    __syncthreads();
#pragma unroll
    for (int u2 = 0; u2 < CQ; ++u2) {   // hidden: thread = channel tid + 256 u2
        const int ch = tid + 256 * u2;
        const int h = ch / (64 * CQ);
        const float* vp = a.v + (size_t)ks * a.ldv + ch;
        float acc = 0.f;
        int j = 0;
        for (; j + 8 <= nk; j += 8) {
            float vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = vp[(size_t)(j + u) * a.ldv];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(sc[h][j + u], vv[u], acc);
        }
        for (; j < nk; ++j) acc = fmaf(sc[h][j], vp[(size_t)j * a.ldv], acc);
        a.out[(size_t)row * a.ldo + ch] = acc;
    }
    // ---- ebar: second pass over this wave's E rows
    float4 acc[NH][CQ];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int c = 0; c < CQ; ++c) acc[h][c] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = wave; j0 < nk; j0 += 16) {
        float4 ev[4][CQ];
#pragma unroll
        for (int u = 0; u < 4; ++u) lde(min(j0 + 4 * u, nk - 1), ev[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 4 * u;
            const float4 p2 = reinterpret_cast<const float4*>(sc2t)[j < nk ? j : 0];
            const float w[4] = {j < nk ? p2.x : 0.f, j < nk ? p2.y : 0.f, j < nk ? p2.z : 0.f, j < nk ? p2.w : 0.f};
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int c = 0; c < CQ; ++c) {
                    acc[h][c].x = fmaf(w[h], ev[u][c].x, acc[h][c].x); acc[h][c].y = fmaf(w[h], ev[u][c].y, acc[h][c].y);
                    acc[h][c].z = fmaf(w[h], ev[u][c].z, acc[h][c].z); acc[h][c].w = fmaf(w[h], ev[u][c].w, acc[h][c].w);
                }
        }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int c = 0; c < CQ; ++c) reinterpret_cast<float4*>(red[wave])[h * (64 * CQ) + lane * CQ + c] = acc[h][c];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int u2 = 0; u2 < CQ; ++u2) {
            const int ch = tid + 256 * u2;
            a.ebar[((size_t)row * NH + h) * C + ch] = (red[0][h * C + ch] + red[1][h * C + ch]) + (red[2][h * C + ch] + red[3][h * C + ch]);
        }
}

template <int NKP, int CQ>   // plain (cross) attention at C = 256 CQ
__global__ __launch_bounds__(256) void mha_plain_wide_kernel(RoitrMha a)
{
    constexpr int NH = 4;
    __shared__ float sc[NH][NKP];
    const int rowi = xcd_block_id(a.q_rows);
    if (rowi >= a.q_rows) return;
    const int row = a.q_row0 + rowi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 4;
    const int cl = a.cloud_of_row[row];
    const int kc = a.partner ? a.partner[cl] : cl;
    const int ks = kc == 0 ? 0 : a.offset[kc - 1], nk = a.offset[kc] - ks;
    float4 qv[CQ];
    ldrow<CQ>(a.q + (size_t)row * a.ldq, lane, qv);
    for (int j0 = wave; j0 < nk; j0 += 16) {   // 4 keys of this wave per trip, loads issued together
        float4 kv[4][CQ];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 4 * u;
            ldrow<CQ>(a.k + (size_t)(ks + (j < nk ? j : nk - 1)) * a.ldk, lane, kv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 4 * u;
            const float s_ = row_allsum(dotq<CQ>(qv, kv[u]));
            if (j < nk && (lane & 15) == 0) sc[hl][j] = s_ * a.scale;
        }
    }
    __syncthreads();
    {   // softmax over the keys: wave = head
        const int h = wave;
        float e1[NKP / 64];
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < NKP / 64; ++u) { const int j = lane + 64 * u; e1[u] = j < nk ? sc[h][j] : -INFINITY; mx = fmaxf(mx, e1[u]); }
        mx = wave_max(mx);
        float sm = 0.f;
#pragma unroll
        for (int u = 0; u < NKP / 64; ++u) { const int j = lane + 64 * u; e1[u] = j < nk ? expf(e1[u] - mx) : 0.f; sm += e1[u]; }
        sm = wave_sum(sm);
#pragma unroll
        for (int u = 0; u < NKP / 64; ++u) { const int j = lane + 64 * u; if (j < nk) sc[h][j] = e1[u] / sm; }
    }
    __syncthreads();
#pragma unroll
    for (int u2 = 0; u2 < CQ; ++u2) {
        const int ch = tid + 256 * u2;
        const int h = ch / (64 * CQ);
        const float* vp = a.v + (size_t)ks * a.ldv + ch;
        float acc = 0.f;
        int j = 0;
        for (; j + 8 <= nk; j += 8) {
            float vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = vp[(size_t)(j + u) * a.ldv];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(sc[h][j + u], vv[u], acc);
        }
        for (; j < nk; ++j) acc = fmaf(sc[h][j], vp[(size_t)j * a.ldv], acc);
        a.out[(size_t)row * a.ldo + ch] = acc;
    }
}

}  // namespace

extern "C" int roitr_geo_indices(int rows, const float* pts, const int* offset, const int* cloud_of_row, const long* eoff,
                                 float sigma_d, float sigma_a, int angle_k, int n_max, float* d_idx, float* a_idx, hipStream_t stream)
{
    if (rows <= 0) return ROITR_OK;
    if (n_max > 1024 || angle_k > 6) return ROITR_ERR_UNSUPPORTED;
    const float factor_a = (float)(180.0 / ((double)sigma_a * 3.14159265358979323846));
    geo_indices_kernel<<<rows, 256, 0, stream>>>(pts, offset, cloud_of_row, eoff, 0.f, sigma_d, factor_a, angle_k, d_idx, a_idx);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_mha(const RoitrMha* a, hipStream_t stream)
{
    if (a->q_rows <= 0) return ROITR_OK;
    const int c = a->C / a->heads;
    if (a->heads > 8 || a->C % a->heads || c % 4 || a->ldk % 4 || a->C % 4) return ROITR_ERR_UNSUPPORTED;
    size_t floats = (size_t)a->C + (a->E ? (size_t)a->heads * a->C : 0) + (size_t)a->heads * a->nk_max * (a->E ? 2 : 1) + 8;
    if (floats * 4 > 150 * 1024) return ROITR_ERR_UNSUPPORTED;
    ROITR_GRANT_LDS(mha_kernel, 150 * 1024);
    roitr_prof_begin(ROITR_PROF_MHA, -1.0, stream);   // bytes: roitr_prof_next_bytes of the caller (0 otherwise)
    {   // factor-2 width (C = 512) and / or E stored in bf16: the wide kernels
        const bool lay = a->heads == 4 && a->ldq % 4 == 0 && a->ldk % 4 == 0 && a->nk_max <= 512;
        if (a->E && !a->partner && lay && (a->C == 512 || (a->C == 256 && a->e_bf16))) {
            const unsigned gr = (unsigned)xcd_grid(a->q_rows);
            if (a->C == 512) { if (a->e_bf16) mha_geo_wide_kernel<2, true><<<gr, 256, 0, stream>>>(*a); else mha_geo_wide_kernel<2, false><<<gr, 256, 0, stream>>>(*a); }
            else mha_geo_wide_kernel<1, true><<<gr, 256, 0, stream>>>(*a);
            roitr_prof_end(ROITR_PROF_MHA, stream);
            ROITR_LAUNCH_CHECK();
            return ROITR_OK;
        }
        if (a->E && a->e_bf16) { roitr_prof_end(ROITR_PROF_MHA, stream); roitr_set_error("roitr_mha: bf16 E needs C = 256 or 512, 4 heads, <= 512 keys", __FILE__, __LINE__); return ROITR_ERR_UNSUPPORTED; }
        if (!a->E && lay && a->C == 512) {
            mha_plain_wide_kernel<512, 2><<<xcd_grid(a->q_rows), 256, 0, stream>>>(*a);
            roitr_prof_end(ROITR_PROF_MHA, stream);
            ROITR_LAUNCH_CHECK();
            return ROITR_OK;
        }
    }
    // self attention over E at the model's width: the single-pass register-resident kernel (nk_max bounds every cloud)
    const bool geo_any = a->E && !a->partner && a->C == 256 && a->heads == 4 && a->ldq % 4 == 0 && a->ldk % 4 == 0;
    const bool geo = geo_any && a->nk_max <= 128;
    if (geo_any && !geo && a->nk_max <= 1024) {
        mha_geo_stream_kernel<<<xcd_grid(a->q_rows), 256, 0, stream>>>(*a);
        roitr_prof_end(ROITR_PROF_MHA, stream);
        ROITR_LAUNCH_CHECK();
        return ROITR_OK;
    }
    const bool plain = !a->E && a->C == 256 && a->heads == 4 && a->ldq % 4 == 0 && a->ldk % 4 == 0 && a->nk_max <= 1024;
    if (plain && a->nk_max <= 128) mha_plain_kernel<128><<<xcd_grid(div_up(a->q_rows, 4)), 256, 0, stream>>>(*a);
    else if (plain) mha_plain_kernel<1024><<<xcd_grid(div_up(a->q_rows, 4)), 256, 0, stream>>>(*a);
    else if (geo && a->nk_max <= 80) mha_geo_kernel<20><<<xcd_grid(a->q_rows), 256, 0, stream>>>(*a);
    else if (geo) mha_geo_kernel<32><<<xcd_grid(a->q_rows), 256, 0, stream>>>(*a);
    else
    mha_kernel<<<xcd_grid(a->q_rows), 256, floats * sizeof(float), stream>>>(*a);
    roitr_prof_end(ROITR_PROF_MHA, stream);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}
