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

namespace {

struct Slot {
    long long key;
    float x, y, z, pad;
};

// Wave-wide max of a signed 64-bit key with DPP row shifts / row broadcasts (VALU-rate cross-lane moves;
// a __shfl_xor butterfly lowers to ds_bpermute, one LDS round trip per step, ~6x the latency on this
// serial critical path).  Lanes without a DPP source keep their own value (old = src, bound_ctrl = 0), which is
// harmless for an idempotent max.  After the row_shr steps lane 15 of every 16-lane row holds the row max,
// row_bcast:15 / :31 fold the rows, lane 63 ends with the wave max.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_max_step(long long v)
{
    const int lo = (int)v, hi = (int)(v >> 32);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const long long w = (long long)(((unsigned long long)(unsigned)hi2 << 32) | (unsigned)lo2);
    return w > v ? w : v;
}

__device__ __forceinline__ long long wave_max_i64(long long v)
{
    v = dpp_max_step<0x111, 0xf>(v);  // row_shr:1
    v = dpp_max_step<0x112, 0xf>(v);  // row_shr:2
    v = dpp_max_step<0x114, 0xf>(v);  // row_shr:4
    v = dpp_max_step<0x118, 0xf>(v);  // row_shr:8
    v = dpp_max_step<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v = dpp_max_step<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    const int lo = __builtin_amdgcn_readlane((int)v, 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// Key = (float bits of d2) << 32 | tie-break.  d2 >= 0 for real points, so the bits order like
// the values; empty register slots carry d2 = -1 (sign bit set => negative key => never wins).
// tie-break = ((1023 - bitrev((k-start) mod bs_ref)) << 21 | (0x1FFFFF - (k-start))) + 1, larger wins.
__device__ __forceinline__ unsigned tie_field(int koff, int bs_ref_mask, int bs_ref_bits)
{
    const unsigned t = (unsigned)(koff & bs_ref_mask);
    const unsigned rev = bs_ref_bits ? (__brev(t) >> (32 - bs_ref_bits)) : 0u;
    return ((1023u - rev) << 21 | (0x1FFFFFu - (unsigned)koff)) + 1u;
}

template <int BLOCK, int PPT>
__global__ __launch_bounds__(BLOCK) void fps_kernel(const float* __restrict__ xyz, const int* __restrict__ offset,
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
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    float px[PPT], py[PPT], pz[PPT], pt[PPT];
    // (k-start) mod bs_ref only depends on j mod 4 because bs_ref <= 4*BLOCK for every dispatch below
    unsigned tba[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        tba[r] = tie_field(tid + r * BLOCK, bs_ref_mask, bs_ref_bits) + (unsigned)(r * BLOCK);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int koff = tid + j * BLOCK;
        if (koff < n) {
            const float* p = xyz + (size_t)(start_n + koff) * 3;
            px[j] = p[0]; py[j] = p[1]; pz[j] = p[2];
            pt[j] = tmp[start_n + koff];
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

    for (int jm = start_m + 1; jm < end_m; ++jm) {
        long long best = -1ll;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float d = sqdist3(px[j], py[j], pz[j], ox, oy, oz);
            const float d2 = fminf(d, pt[j]);
            pt[j] = d2;
            const unsigned tb = tba[j & 3] - (unsigned)(j * BLOCK);
            const long long key = (long long)((unsigned long long)__float_as_uint(d2) << 32 | tb);
            best = key > best ? key : best;
        }
        const long long wbest = wave_max_i64(best);
        Slot* buf = slots[jm & 1];
        if (best == wbest && (wbest >= 0 || lane == 0)) {
            // this lane owns the wave's winner: recover its register slot, publish key + coordinates
            float wx = 0.f, wy = 0.f, wz = 0.f;
            const unsigned wtb = (unsigned)wbest;
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const bool hit = (tba[j & 3] - (unsigned)(j * BLOCK)) == wtb;
                wx = hit ? px[j] : wx; wy = hit ? py[j] : wy; wz = hit ? pz[j] : wz;
            }
            buf[wave].key = wbest; buf[wave].x = wx; buf[wave].y = wy; buf[wave].z = wz;
        }
        __syncthreads();
        long long gbest = buf[0].key;
        int gw = 0;
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const long long k = buf[w].key;
            if (k > gbest) { gbest = k; gw = w; }
        }
        int old = start_n;
        if (gbest >= 0) {
            old = start_n + (int)(0x1FFFFFu - (((unsigned)gbest - 1u) & 0x1FFFFFu));
            ox = buf[gw].x; oy = buf[gw].y; oz = buf[gw].z;
        } else if (n > 0) {
            const float* p = xyz + (size_t)start_n * 3;
            ox = p[0]; oy = p[1]; oz = p[2];
        }
        if (tid == 0) idx[jm] = old;
    }

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

// cuda_utils.h:11-14: the block size the reference would launch, same double-precision formula
int ref_block_size(int n)
{
    if (n < 1) return 1;
    const int pow_2 = (int)(std::log((double)n) / std::log(2.0));
    return std::max(std::min(1 << pow_2, 1024), 1);
}

}  // namespace

extern "C" int roitr_furthestsampling(int b, int n_max, const float* xyz, const int* offset, const int* new_offset,
                                      float* tmp, int* idx, hipStream_t stream)
{
    if (b <= 0) return ROITR_OK;
    if (n_max >= (1 << 21)) return ROITR_ERR_UNSUPPORTED;
    const int mask = ref_block_size(n_max) - 1;
    int bits = 0;
    while ((1 << bits) <= mask) ++bits;
#define FPS_CASE(BLK, P)                                                                              \
    if (n_max <= (BLK) * (P)) {                                                                       \
        roitr_prof_begin(ROITR_PROF_FPS, -1.0, stream);                                               \
        fps_kernel<BLK, P><<<b, BLK, 0, stream>>>(xyz, offset, new_offset, tmp, idx, mask, bits);     \
        roitr_prof_end(ROITR_PROF_FPS, stream);                                                       \
        ROITR_LAUNCH_CHECK();                                                                         \
        return ROITR_OK;                                                                              \
    }
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
    FPS_CASE(1024, 16)
    FPS_CASE(1024, 24)
#undef FPS_CASE
    fps_stream_kernel<1024><<<b, 1024, 0, stream>>>(xyz, offset, new_offset, tmp, idx, mask, bits);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

// Exact drop-in for sampling_cuda_kernel.h:9-17 (void return, legacy default stream).
extern "C" void furthestsampling_cuda_launcher(int b, int n, const float* xyz, const int* offset, const int* new_offset,
                                               float* tmp, int* idx)
{
    (void)roitr_furthestsampling(b, n, xyz, offset, new_offset, tmp, idx, nullptr);
}
