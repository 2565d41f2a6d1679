// Furthest-point sampling for gfx950.
//
// Replaces furthestsampling_cuda_launcher (reference cpp_wrappers/pointops/src/sampling/
// sampling_cuda_kernel.cu:15-170) behind the same C signature.  Not a translation of that kernel:
//   * one workgroup per cloud, but the cloud's xyz AND its running min-distances live in VGPRs for
//     the whole sampling run (PPT points per lane); HBM is touched once on entry and once on exit
//     (algorithmic bytes 12n + 4m + 8n for the in/out `tmp`), never inside the m-iteration loop;
//   * 4 waves per workgroup (one per SIMD; 8 when a call has few clouds) instead of 16; the per-iteration arg-max is two chained
//     32-bit maxima -- the largest distance, then the largest tie word among the points that attain it -- each a 6-step DPP reduction
//     of ONE instruction per step, then one 16-byte LDS slot per wave, ONE barrier per iteration (double-buffered slots) and a 2- or 3-step
//     DPP fold of the slots, versus the reference's 10-step shared-memory tree with 11 barriers;
//   * two points share a register pair: the distance update is v_pk_add / v_pk_mul / v_pk_fma on both at once (round 4);
//   * the winner's coordinates come from an LDS copy of the cloud (few clouds per call) or from L2 (large batches: the copy's 76 KB
//     per cloud would starve the feature path's workgroups), never from a dependent HBM miss.
//
// Bit-exactness: the reference's result depends on its launch shape -- thread `tid` of a
// `bs`-thread block (bs = opt_n_threads(n), cuda_utils.h:11-14) scans k = start+tid, +bs, ...
// keeping the first strict maximum (l.49-59); the shared-memory tree (l.64-123) merges slot t with
// slot t+s for s = bs/2 ... 1 and keeps the LOWER SLOT on equal values (l.5-10).  Because slot t
// already holds the winner of {t, t+bs/2} when it meets slot t+bs/4, the tournament's tie order is
// not "lowest tid": the group with tid bit0 = 0 beats bit0 = 1, inside it bit1 = 0 beats bit1 = 1, ...
// i.e. the total order is "max d, then min BITREVERSE(tid) over log2(bs) bits, then min k".  It is
// folded into the low 32 bits of the reduction key, so any reduction shape reproduces the reference.
#include "common.h"
#include "prof.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace {

struct Slot {
    long long key;
    float x, y, z, pad;
};

// ---- cross-lane reductions as FUSED DPP instructions: one VALU op per butterfly step (`v_max_f32_dpp v, v, v row_shr:1`).  The
// compiler's lowering of update_dpp + fmaxf is four (copy, v_mov_dpp, a NaN-quieting v_max v,v,v, v_max), and every instruction of
// this loop is on the serial critical path of the sampling chain (a wave issues one instruction per 4 cycles whatever its kind).
// Lanes without a DPP source (bound_ctrl 0) and rows outside row_mask keep their value -- harmless for max, and for the sums below
// the lane that is read back always has all its sources.  `s_nop 1`: the two wait states between a VALU write and a DPP read of
// the same VGPR, which nobody inserts inside an asm statement.  After row_shr 1/2/4/8 lane 15 of every 16-lane row holds the row's
// result; row_bcast:15 / :31 fold the rows; lane 63 ends with the wave's.
#define ROITR_DPP(op, v, ctrl) asm("s_nop 1\n\t" op " %0, %0, %0 " ctrl : "+v"(v))
#define ROITR_DPP_WAVE(op, v)                                    \
    ROITR_DPP(op, v, "row_shr:1 row_mask:0xf bank_mask:0xf");    \
    ROITR_DPP(op, v, "row_shr:2 row_mask:0xf bank_mask:0xf");    \
    ROITR_DPP(op, v, "row_shr:4 row_mask:0xf bank_mask:0xf");    \
    ROITR_DPP(op, v, "row_shr:8 row_mask:0xf bank_mask:0xf");    \
    ROITR_DPP(op, v, "row_bcast:15 row_mask:0xa bank_mask:0xf"); \
    ROITR_DPP(op, v, "row_bcast:31 row_mask:0xc bank_mask:0xf"); \
    asm("s_nop 1" ::: )
__device__ __forceinline__ float wave_fmax(float v)
{
    ROITR_DPP_WAVE("v_max_f32_dpp", v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ unsigned wave_umax(unsigned v)
{
    ROITR_DPP_WAVE("v_max_u32_dpp", v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// the same over groups of G = 2, 4, 8 or 16 consecutive lanes: lane G-1 of the first group ends with the group's result
template <int G, typename T>
__device__ __forceinline__ T group_reduce_read(T v, const int kind)   // kind 0: fmax, 1: umax, 2: uadd
{
#define ROITR_DPP_G(op)                                                          \
    if (G > 1) ROITR_DPP(op, v, "row_shr:1 row_mask:0xf bank_mask:0xf");         \
    if (G > 2) ROITR_DPP(op, v, "row_shr:2 row_mask:0xf bank_mask:0xf");         \
    if (G > 4) ROITR_DPP(op, v, "row_shr:4 row_mask:0xf bank_mask:0xf");         \
    if (G > 8) ROITR_DPP(op, v, "row_shr:8 row_mask:0xf bank_mask:0xf");
    if (kind == 0) { ROITR_DPP_G("v_max_f32_dpp") } else if (kind == 1) { ROITR_DPP_G("v_max_u32_dpp") } else { ROITR_DPP_G("v_add_u32_dpp") }
#undef ROITR_DPP_G
    asm("s_nop 1" ::: );
    int bits;
    __builtin_memcpy(&bits, &v, 4);
    bits = __builtin_amdgcn_readlane(bits, G - 1);
    T r;
    __builtin_memcpy(&r, &bits, 4);
    return r;
}
// v_min_f32 / v_max3 as written: fminf / fmaxf on a loop-carried register put a NaN-quieting `v_max_f32 v, v, v` in front of every use
__device__ __forceinline__ float vmin_f32(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3_f32(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned vmax3_u32(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ long long wave_max_i64(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}

// The reference order "max d, then min bitrev((k-start) mod bs_ref), then min k" is evaluated as two chained
// 32-bit maxima instead of one 64-bit key: first the maximum distance (v_max_f32, no serial compare/select chain),
// then, among the elements that attain it, the maximum tie-break word
//     tb = ((1023 - bitrev((k-start) mod bs_ref)) << 21 | (0x1FFFFF - (k-start))) + 1      (larger = preferred).
// Distances are >= 0; register slots that hold no point carry d2 = -1 and can never attain a maximum >= 0.
__device__ __forceinline__ unsigned tie_field(int koff, int bs_ref_mask, int bs_ref_bits)
{
    const unsigned t = (unsigned)(koff & bs_ref_mask);
    const unsigned rev = bs_ref_bits ? (__brev(t) >> (32 - bs_ref_bits)) : 0u;
    return ((1023u - rev) << 21 | (0x1FFFFFu - (unsigned)koff)) + 1u;
}

constexpr int FPS_IDX_CAP = 4096;    // selected indices parked in LDS (written out once at the end)
constexpr int FPS_PTS_CAP = 8192;    // clouds up to this size keep an xyz copy in LDS for the winner lookup

// Hierarchy shortcut (round 3).  The next level samples the picks of this level IN PICK ORDER from the same first point, so
// as long as every arg-max of this level's first m' iterations was attained by exactly ONE point, FPS on those picks returns
// their first m' positions: the running minimum distances of the picked points are the same numbers (same coordinates, same
// arithmetic), the maximum over the subset is the maximum over the whole cloud and it sits at the same, unique, point.  A tie
// is the only place where the reference's block-tournament order (which depends on the cloud size) could choose differently.
// `tie_out[cloud]` = first pick index at which the maximum was shared (`track`: none among the first `track` picks);
// `prev_tie` != null: this launch samples the previous level's picks -- a cloud whose prev_tie covers all m picks writes the
// prefix and leaves (passing the value on for the level after it), every other cloud runs the real thing.
template <int BLOCK, int PPT>
__global__ __launch_bounds__(BLOCK) void fps_kernel(const float* __restrict__ xyz, const int* __restrict__ offset,
                                                    const int* __restrict__ new_offset, float* __restrict__ tmp,
                                                    int* __restrict__ idx, int bs_ref_mask, int bs_ref_bits, int lds_pts,
                                                    const int* __restrict__ prev_tie, int* __restrict__ tie_out, int track_div, int b)
{
    constexpr int NW = BLOCK / 64;
    constexpr int NP = PPT / 2;                                            // two points per packed-fp32 register pair
    static_assert(PPT % 2 == 0 && NW <= 16, "points are held in pairs; the cross-wave stage is one DPP group");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* slots = reinterpret_cast<float4*>(smem);                       // [2][NW] : (max d2, tie word, points attaining it, -)
    int* sidx = reinterpret_cast<int*>(smem + 2 * NW * sizeof(float4));    // [FPS_IDX_CAP]
    // xyz copy for the winner lookup, three planes of lds_pts floats (12 B per point: two 5000-point clouds share a CU)
    float* spts = reinterpret_cast<float*>(smem + 2 * NW * sizeof(float4) + FPS_IDX_CAP * sizeof(int));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // a workgroup takes clouds blockIdx.x, + gridDim.x, ...: the launcher bounds the grid in large batches so that the sampling
    // chains occupy a bounded share of every CU's registers (see roitr_furthestsampling_ex)
  for (int bid = blockIdx.x; bid < b; bid += gridDim.x) {
    const int start_n = bid == 0 ? 0 : offset[bid - 1];
    const int end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1];
    const int end_m = new_offset[bid];
    const int n = end_n - start_n;
    const bool pts_in_lds = n <= lds_pts;
    if (prev_tie) {
        const int pt_ = prev_tie[bid];
        if (pt_ >= end_m - start_m) {                       // block-uniform: the picks are the cloud's first m points
            for (int j = tid; j < end_m - start_m; j += BLOCK) idx[start_m + j] = start_n + j;
            if (tie_out && tid == 0) tie_out[bid] = pt_;
            continue;
        }
    }
    // picks whose arg-max uniqueness is recorded: the next level keeps (m / track_div) of this level's m picks
    const int track = (tie_out && track_div > 0) ? (end_m - start_m) / track_div : 0;
    int first_tie = 0x7fffffff;

    // register slot j of thread tid holds point tid + j * BLOCK; slots 2p and 2p+1 share a register pair so that the three
    // differences, the square and the two fused multiply-adds of a distance are v_pk_*_f32 instructions on two points at once
    // (same IEEE operations per element as sqdist3: bit-identical, 6 instructions per two points instead of 12)
    f32x2 px[NP], py[NP], pz[NP];
    float pt[PPT];
    // (k-start) mod bs_ref only depends on j mod 4 because bs_ref <= 4*BLOCK for every dispatch below
    unsigned tba[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tba[r] = tie_field(tid + r * BLOCK, bs_ref_mask, bs_ref_bits) + (unsigned)(r * BLOCK);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int koff = tid + j * BLOCK;
        float x = 0.f, y = 0.f, z = 0.f;
        pt[j] = -1.f;
        if (koff < n) {
            const float* p = xyz + (size_t)(start_n + koff) * 3;
            x = p[0]; y = p[1]; z = p[2];
            pt[j] = tmp[start_n + koff];
            if (pts_in_lds) { spts[koff] = x; spts[lds_pts + koff] = y; spts[2 * lds_pts + koff] = z; }
        }
        px[j >> 1][j & 1] = x; py[j >> 1][j & 1] = y; pz[j >> 1][j & 1] = z;
    }

    if (tid == 0 && start_m < end_m) idx[start_m] = start_n;
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (n > 0) {
        const float* p = xyz + (size_t)start_n * 3;  // `old` starts at the segment's first point
        ox = p[0]; oy = p[1]; oz = p[2];
    }
    __syncthreads();

    for (int jm = start_m + 1; jm < end_m; ++jm) {
        float dmax = -1.f;
        const f32x2 o2x = {ox, ox}, o2y = {oy, oy}, o2z = {oz, oz};
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const f32x2 dx = px[p] - o2x, dy = py[p] - o2y, dz = pz[p] - o2z;
            const f32x2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
            pt[2 * p] = vmin_f32(d[0], pt[2 * p]);
            pt[2 * p + 1] = vmin_f32(d[1], pt[2 * p + 1]);
            dmax = vmax3_f32(dmax, pt[2 * p], pt[2 * p + 1]);
        }
        const float wd = wave_fmax(dmax);
        const bool tracked = jm - start_m < track;           // block-uniform
        unsigned btb = 0u;
        unsigned long long at[PPT];                          // lanes whose slot j attains the wave maximum: the compare's SGPR pair
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            at[2 * p] = __builtin_amdgcn_ballot_w64(pt[2 * p] == wd);
            at[2 * p + 1] = __builtin_amdgcn_ballot_w64(pt[2 * p + 1] == wd);
            const unsigned c0 = pt[2 * p] == wd ? tba[(2 * p) & 3] - (unsigned)(2 * p * BLOCK) : 0u;
            const unsigned c1 = pt[2 * p + 1] == wd ? tba[(2 * p + 1) & 3] - (unsigned)((2 * p + 1) * BLOCK) : 0u;
            btb = vmax3_u32(btb, c0, c1);
        }
        int wn = 0;                                          // points of this wave that attain its maximum (scalar unit, tracked picks only)
        if (tracked) {
#pragma unroll
            for (int j = 0; j < PPT; ++j) wn += __builtin_popcountll(at[j]);
        }
        const unsigned wtb = wave_umax(btb);
        float4* buf = slots + (jm & 1) * NW;
        if (lane == 0) buf[wave] = make_float4(wd, __uint_as_float(wtb), __int_as_float(wn), 0.f);
        // LDS-only barrier: wait for this wave's LDS write, not for outstanding global traffic
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // cross-wave stage: lane l takes the slot of wave l mod NW (one ds_read_b128), the NW slots are folded with DPP steps
        const float4 sl = buf[lane & (NW - 1)];
        const float gd = group_reduce_read<NW>(sl.x, 0);
        const bool top = sl.x == gd;
        const unsigned gtb = group_reduce_read<NW>(top ? __float_as_uint(sl.y) : 0u, 1);
        if (tracked) {
            const unsigned tot = group_reduce_read<NW>(top ? __float_as_uint(sl.z) : 0u, 2);
            if (tot != 1u && first_tie == 0x7fffffff) first_tie = jm - start_m;
        }
        int old = start_n;
        if (gd >= 0.f) old = start_n + (int)(0x1FFFFFu - ((gtb - 1u) & 0x1FFFFFu));
        if (pts_in_lds) {
            const int oi = old - start_n;
            ox = spts[oi]; oy = spts[lds_pts + oi]; oz = spts[2 * lds_pts + oi];
        } else if (n > 0) {
            const float* p = xyz + (size_t)old * 3;
            ox = p[0]; oy = p[1]; oz = p[2];
        }
        if (tid == 0) {
            if (jm - start_m < FPS_IDX_CAP) sidx[jm - start_m] = old;
            else idx[jm] = old;
        }
    }
    __syncthreads();
    for (int j = 1 + tid; j < end_m - start_m && j < FPS_IDX_CAP; j += BLOCK) idx[start_m + j] = sidx[j];
    // min(first tie, tracked picks): the next level's shortcut test `prev_tie >= m'` can then only pass for m' <= track, i.e. for
    // picks whose uniqueness WAS recorded, whatever divisor the caller of the next level uses (0: nothing tracked)
    if (tie_out && tid == 0) tie_out[bid] = track > 0 ? (first_tie < track ? first_tie : track) : 0;

#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int koff = tid + j * BLOCK;
        if (koff < n) tmp[start_n + koff] = pt[j];
    }
    __syncthreads();   // the next cloud reuses the slots, the pick list and the xyz copy
  }
}

// Clouds beyond the register-resident limit: same key scheme, `tmp` streamed through L2.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void fps_stream_kernel(const float* __restrict__ xyz, const int* __restrict__ offset,
                                                           const int* __restrict__ new_offset, float* __restrict__ tmp,
                                                           int* __restrict__ idx, int bs_ref_mask, int bs_ref_bits)
{
    constexpr int NW = BLOCK / 64;
    __shared__ Slot slots[2][NW];
    const int bid = blockIdx.x;
    const int start_n = bid == 0 ? 0 : offset[bid - 1];
    const int end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1];
    const int end_m = new_offset[bid];
    const int n = end_n - start_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0 && start_m < end_m) idx[start_m] = start_n;
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (n > 0) { ox = xyz[(size_t)start_n * 3]; oy = xyz[(size_t)start_n * 3 + 1]; oz = xyz[(size_t)start_n * 3 + 2]; }
    for (int jm = start_m + 1; jm < end_m; ++jm) {
        long long best = -1ll;
        for (int koff = tid; koff < n; koff += BLOCK) {
            const float* p = xyz + (size_t)(start_n + koff) * 3;
            const float d2 = fminf(sqdist3(p[0], p[1], p[2], ox, oy, oz), tmp[start_n + koff]);
            tmp[start_n + koff] = d2;
            const unsigned t = tie_field(koff, bs_ref_mask, bs_ref_bits);
            const long long key = (long long)((unsigned long long)__float_as_uint(d2) << 32 | t);
            best = key > best ? key : best;
        }
        const long long wbest = wave_max_i64(best);
        Slot* buf = slots[jm & 1];
        if (lane == 0) buf[wave].key = wbest;
        __syncthreads();
        long long gbest = buf[0].key;
        for (int w = 1; w < NW; ++w) gbest = buf[w].key > gbest ? buf[w].key : gbest;
        int old = start_n;
        if (gbest >= 0) old = start_n + (int)(0x1FFFFFu - (((unsigned)gbest - 1u) & 0x1FFFFFu));
        if (n > 0) { ox = xyz[(size_t)old * 3]; oy = xyz[(size_t)old * 3 + 1]; oz = xyz[(size_t)old * 3 + 2]; }
        if (tid == 0) idx[jm] = old;
    }
}

size_t fps_lds_bytes(int block, int lds_pts)
{
    return (size_t)2 * (block / 64) * sizeof(float4) + FPS_IDX_CAP * sizeof(int) + (size_t)lds_pts * 3 * sizeof(float);
}

// cuda_utils.h:11-14: the block size the reference would launch, same double-precision formula
int ref_block_size(int n)
{
    if (n < 1) return 1;
    const int pow_2 = (int)(std::log((double)n) / std::log(2.0));
    return std::max(std::min(1 << pow_2, 1024), 1);
}

}  // namespace

extern "C" int roitr_furthestsampling_ex(int b, int n_max, const float* xyz, const int* offset, const int* new_offset, float* tmp, int* idx,
                                         const int* prev_tie, int* tie_out, int track_div, hipStream_t stream);
extern "C" int roitr_furthestsampling(int b, int n_max, const float* xyz, const int* offset, const int* new_offset,
                                      float* tmp, int* idx, hipStream_t stream)
{
    return roitr_furthestsampling_ex(b, n_max, xyz, offset, new_offset, tmp, idx, nullptr, nullptr, 0, stream);
}

/* The sampling chain of a hierarchy (see fps_kernel): tie_out (b ints, device) receives per cloud the first pick index whose
 * arg-max was shared among this level's first (m / track_div) picks; prev_tie = the tie_out of the level whose PICKS (in pick
 * order) this call samples -- clouds it covers are answered with the prefix 0 .. m-1 without running the chain.  Both optional. */
extern "C" int roitr_furthestsampling_ex(int b, int n_max, const float* xyz, const int* offset, const int* new_offset, float* tmp, int* idx,
                                         const int* prev_tie, int* tie_out, int track_div, hipStream_t stream)
{
    if (b <= 0) return ROITR_OK;
    if (n_max >= (1 << 21)) return ROITR_ERR_UNSUPPORTED;
    const int mask = ref_block_size(n_max) - 1;
    int bits = 0;
    while ((1 << bits) <= mask) ++bits;
    // the LDS copy of xyz (winner lookup without a memory round trip) is what a FEW clouds on an empty chip want; in a large batch it
    // only takes LDS away from everybody else: 76 KB per cloud = two clouds per CU and nothing left for the feature path's
    // workgroups beside them
    constexpr int lds_max_b = 64;
    const int lds_pts = (n_max <= FPS_PTS_CAP && b <= lds_max_b) ? n_max : 0;
    // few clouds (the one-pair-per-call mode): the chain of m dependent arg-max iterations is the critical path of the whole
    // forward and the chip is empty -- 8 waves per cloud halve the per-lane work of an iteration (4.36 vs 4.48 ms per pair;
    // 16 waves: 5.03, the cross-wave stage grows faster than the lane work shrinks).  Same indices for every block size.
    // (Round 4 measured 8-wave workgroups for large batches and a cap on the resident clouds per CU as well: no effect, removed.)
    const int forced = n_max <= 512 * 16 && b <= 16 ? 512 : 0;
    const int nblk = b;
#define FPS_CASE(BLK, P)                                                                              \
    if (n_max <= (BLK) * (P) && (forced == 0 || forced == (BLK))) {                                   \
        ROITR_GRANT_LDS((fps_kernel<BLK, P>), fps_lds_bytes(BLK, FPS_PTS_CAP));                       \
        roitr_prof_begin(ROITR_PROF_FPS, -1.0, stream);                                               \
        fps_kernel<BLK, P><<<nblk, BLK, fps_lds_bytes(BLK, lds_pts), stream>>>(xyz, offset, new_offset, tmp, idx, mask, bits, lds_pts, prev_tie, tie_out, \
                                                                               track_div, b);                                       \
        roitr_prof_end(ROITR_PROF_FPS, stream);                                                       \
        ROITR_LAUNCH_CHECK();                                                                         \
        return ROITR_OK;                                                                              \
    }
    if (forced == 0) {
        FPS_CASE(64, 2)
        FPS_CASE(256, 2)
        FPS_CASE(256, 4)
        FPS_CASE(256, 8)
        FPS_CASE(256, 12)
        FPS_CASE(256, 16)
        FPS_CASE(256, 20)
        FPS_CASE(256, 24)
        FPS_CASE(256, 32)
        FPS_CASE(512, 24)
        FPS_CASE(512, 32)
    } else {
        FPS_CASE(512, 2)
        FPS_CASE(512, 4)
        FPS_CASE(512, 10)
        FPS_CASE(512, 16)
    }
#undef FPS_CASE
    // Measured and dropped for 16 k .. 30 k points: coordinates in registers (512 threads x 60 points) with the running
    // min-distances in LDS -- the 180 coordinate registers spill (450 dwords) and the forward at N = 30000 got slower
    // (101 vs 75 ms for 2 pairs) than with the L2-streaming kernel below (6.9 us per iteration).
    fps_stream_kernel<1024><<<b, 1024, 0, stream>>>(xyz, offset, new_offset, tmp, idx, mask, bits);
    ROITR_LAUNCH_CHECK();
    if (tie_out) ROITR_HIP(hipMemsetAsync(tie_out, 0, sizeof(int) * (size_t)b, stream));   // untracked: the next level runs its own chain
    return ROITR_OK;
}

// Exact drop-in for sampling_cuda_kernel.h:9-17 (void return, legacy default stream).
extern "C" void furthestsampling_cuda_launcher(int b, int n, const float* xyz, const int* offset, const int* new_offset,
                                               float* tmp, int* idx)
{
    (void)roitr_furthestsampling(b, n, xyz, offset, new_offset, tmp, idx, nullptr);
}
