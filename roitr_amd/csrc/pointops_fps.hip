// Furthest-point sampling for gfx950.
//
// Replaces furthestsampling_cuda_launcher (reference cpp_wrappers/pointops/src/sampling/
// sampling_cuda_kernel.cu:15-170) behind the same C signature.  Not a translation of that kernel:
//   * one workgroup per cloud, but the cloud's xyz AND its running min-distances live in VGPRs for
//     the whole sampling run (PPT points per lane); HBM is touched once on entry and once on exit
//     (algorithmic bytes 12n + 4m + 8n for the in/out `tmp`), never inside the m-iteration loop;
//   * 4 waves per workgroup (one per SIMD) instead of 16: the per-iteration arg-max is a 64-bit
//     key max -- 6 wave64 butterfly steps + one LDS slot per wave + ONE barrier per iteration
//     (double-buffered slots), versus the reference's 10-step shared-memory tree with 11 barriers;
//   * the winner's coordinates ride along with the wave winner through LDS, so the next
//     iteration never goes back to memory for xyz[old].
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

// ---- cross-lane maxima with DPP row shifts / row broadcasts (VALU-rate; a __shfl_xor butterfly lowers to
// ds_bpermute = one LDS round trip per step on this serial critical path).  Lanes without a DPP source keep
// their own value (old = src, bound_ctrl = 0), harmless for an idempotent max.  After the row_shr steps lane 15
// of every 16-lane row holds the row max; row_bcast:15 / :31 fold the rows; lane 63 ends with the wave max.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_fmax(float v)
{
    const int i = __float_as_int(v);
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xf, false)));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umax(unsigned v)
{
    const unsigned w = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
    return w > v ? w : v;
}
__device__ __forceinline__ float wave_fmax(float v)
{
    v = dpp_fmax<0x111, 0xf>(v); v = dpp_fmax<0x112, 0xf>(v); v = dpp_fmax<0x114, 0xf>(v); v = dpp_fmax<0x118, 0xf>(v);
    v = dpp_fmax<0x142, 0xa>(v); v = dpp_fmax<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ unsigned wave_umax(unsigned v)
{
    v = dpp_umax<0x111, 0xf>(v); v = dpp_umax<0x112, 0xf>(v); v = dpp_umax<0x114, 0xf>(v); v = dpp_umax<0x118, 0xf>(v);
    v = dpp_umax<0x142, 0xa>(v); v = dpp_umax<0x143, 0xc>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
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
                                                    const int* __restrict__ prev_tie, int* __restrict__ tie_out, int track_div)
{
    constexpr int NW = BLOCK / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* slots = reinterpret_cast<float2*>(smem);                       // [2][NW] : (max d2, tie word as float bits)
    __shared__ int wcnt[2][NW];                                            // points attaining the wave maximum (tracked picks only)
    int* sidx = reinterpret_cast<int*>(smem + 2 * NW * sizeof(float2));    // [FPS_IDX_CAP]
    // xyz copy for the winner lookup, three planes of lds_pts floats (12 B per point: two 5000-point clouds share a CU)
    float* spts = reinterpret_cast<float*>(smem + 2 * NW * sizeof(float2) + FPS_IDX_CAP * sizeof(int));

    const int bid = blockIdx.x;
    const int start_n = bid == 0 ? 0 : offset[bid - 1];
    const int end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1];
    const int end_m = new_offset[bid];
    const int n = end_n - start_n;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const bool pts_in_lds = n <= lds_pts;
    if (prev_tie) {
        const int pt_ = prev_tie[bid];
        if (pt_ >= end_m - start_m) {                       // block-uniform: the picks are the cloud's first m points
            for (int j = tid; j < end_m - start_m; j += BLOCK) idx[start_m + j] = start_n + j;
            if (tie_out && tid == 0) tie_out[bid] = pt_;
            return;
        }
    }
    // picks whose arg-max uniqueness is recorded: the next level keeps (m / track_div) of this level's m picks
    const int track = (tie_out && track_div > 0) ? (end_m - start_m) / track_div : 0;
    int first_tie = 0x7fffffff;

    float px[PPT], py[PPT], pz[PPT], pt[PPT];
    // (k-start) mod bs_ref only depends on j mod 4 because bs_ref <= 4*BLOCK for every dispatch below
    unsigned tba[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tba[r] = tie_field(tid + r * BLOCK, bs_ref_mask, bs_ref_bits) + (unsigned)(r * BLOCK);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int koff = tid + j * BLOCK;
        if (koff < n) {
            const float* p = xyz + (size_t)(start_n + koff) * 3;
            px[j] = p[0]; py[j] = p[1]; pz[j] = p[2];
            pt[j] = tmp[start_n + koff];
            if (pts_in_lds) { spts[koff] = px[j]; spts[lds_pts + koff] = py[j]; spts[2 * lds_pts + koff] = pz[j]; }
        } else {
            px[j] = py[j] = pz[j] = 0.f;
            pt[j] = -1.f;
        }
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
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float d2 = fminf(sqdist3(px[j], py[j], pz[j], ox, oy, oz), pt[j]);
            pt[j] = d2;
            dmax = fmaxf(dmax, d2);
        }
        const float wd = wave_fmax(dmax);
        const bool tracked = jm - start_m < track;           // block-uniform
        unsigned btb = 0u;
        int nat = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const unsigned tb = tba[j & 3] - (unsigned)(j * BLOCK);
            const bool at = pt[j] == wd;
            const unsigned c = at ? tb : 0u;
            btb = c > btb ? c : btb;
            if (tracked) nat += at ? 1 : 0;
        }
        const unsigned wtb = wave_umax(btb);
        float2* buf = slots + (jm & 1) * NW;
        if (tracked) {
            const int wn = (int)wave_sum((float)nat);          // <= 64 * PPT: exact in fp32
            if (lane == 0) wcnt[jm & 1][wave] = wn;
        }
        if (lane == 0) buf[wave] = make_float2(wd, __uint_as_float(wtb));
        // LDS-only barrier: wait for this wave's LDS write, not for outstanding global traffic
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float gd = -1.f;
        float2 sl[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) { sl[w] = buf[w]; gd = fmaxf(gd, sl[w].x); }
        unsigned gtb = 0u;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const unsigned c = sl[w].x == gd ? __float_as_uint(sl[w].y) : 0u;
            gtb = c > gtb ? c : gtb;
        }
        if (tracked) {
            int tot = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += sl[w].x == gd ? wcnt[jm & 1][w] : 0;
            if (tot != 1 && first_tie == 0x7fffffff) first_tie = jm - start_m;
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
    return (size_t)2 * (block / 64) * sizeof(float2) + FPS_IDX_CAP * sizeof(int) + (size_t)lds_pts * 3 * sizeof(float);
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
    // workgroups beside them (ROITR_FPS_LDS_MAX_CLOUDS: experiment switch)
    static const int lds_max_b = getenv("ROITR_FPS_LDS_MAX_CLOUDS") ? atoi(getenv("ROITR_FPS_LDS_MAX_CLOUDS")) : 64;
    const int lds_pts = (n_max <= FPS_PTS_CAP && b <= lds_max_b) ? n_max : 0;
    // few clouds (the one-pair-per-call mode): the chain of m dependent arg-max iterations is the critical path of the whole
    // forward and the chip is empty -- 8 waves per cloud halve the per-lane work of an iteration (4.36 vs 4.48 ms per pair;
    // 16 waves: 5.03, the cross-wave stage grows faster than the lane work shrinks).  Same indices for every block size.
    const int forced = b <= 16 && n_max <= 512 * 16 ? 512 : 0;
#define FPS_CASE(BLK, P)                                                                              \
    if (n_max <= (BLK) * (P) && (forced == 0 || forced == (BLK))) {                                   \
        ROITR_GRANT_LDS((fps_kernel<BLK, P>), fps_lds_bytes(BLK, FPS_PTS_CAP));                       \
        roitr_prof_begin(ROITR_PROF_FPS, -1.0, stream);                                               \
        fps_kernel<BLK, P><<<b, BLK, fps_lds_bytes(BLK, lds_pts), stream>>>(xyz, offset, new_offset, tmp, idx, mask, bits, lds_pts, prev_tie, tie_out, \
                                                                            track_div);                                             \
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
