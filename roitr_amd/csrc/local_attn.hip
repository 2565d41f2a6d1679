// Local PPF attention over the K grouped neighbours of every node.
//
// Reference: LocalRPEMultiHeadAttention.forward, model/transformer/attention.py:152-200, with the
// positional branch folded algebraically (no nonlinearity sits between PPFStructualEmbedding.proj,
// positional_encoding.py:79, and proj_p / proj_vp, attention.py:169-170):
//     p  = Wp (We ppf + be) + bp  = Wpe ppf + bpe          (Wpe: H x 4)
//     vp = Wvp(We ppf + be) + bvp = Wvpe ppf + bvpe
//     q_h . p_hk      = (Wpe_h^T q_h) . ppf_k + q_h . bpe_h   = qp[h][0:4] . ppf_k + qp[h][4]
//     sum_k a_hk vp_hk = Wvpe_h (sum_k a_hk ppf_k) + bvpe_h    (softmax rows sum to 1)
// so the (M,K,H) tensors p and vp are never formed: the reference spends 2*M*K*H^2 MACs there
// (the dominant dense cost of the encoder); here it is 5 extra GEMM columns
// per head (qp, produced together with q) and 5 FMAs per output channel.
//
// One wave per node.  Phase 1: lane = (head, neighbour) computes its score with 16-byte loads of the
// gathered key row slice, softmax over the neighbour axis is a butterfly inside the head's lane
// group.  Phase 2: lane = channel; the value rows are read fully coalesced.  Gathered k/v rows are
// never materialised (the reference builds (M,K,H) copies with fancy indexing, attention.py:174-175).
#include "common.h"
#include "prof.h"
#include "roitr_engine.h"

namespace {

// K = neighbours per node (8 / 16), HV = H / 64 (channels per lane in phase 2).  Memory round trips per node are the
// critical path (72 % of wave time was s_waitcnt in the first version), so everything that only depends on the node
// id is requested up front (neighbour indices, q row, ppf), and everything that only depends on the neighbour
// indices -- the key slices AND the value rows -- is requested together: two dependent round trips instead of four.
template <int K, int HV>
__global__ __launch_bounds__(256) void local_attn_kernel(RoitrLocalAttn a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // nodes are visited in the grid's cell order when one is given: consecutive waves then gather overlapping
    // neighbour sets, so a k/v row is re-used from L2 instead of being fetched once per referencing node
    const int slot = blockIdx.x * 4 + wave;
    const bool valid = slot < a.M;  // whole-wave predicate (no early exit: block barriers below)
    const int node = (a.node_order && valid) ? __float_as_int(reinterpret_cast<const float4*>(a.node_order)[slot].w) : (valid ? slot : 0);
    constexpr int H = 64 * HV;
    const int NH = a.heads, c = H / NH;
    float* qs = smem + (size_t)wave * (H + 5 * NH + 64 + 4 * NH + 64);  // q row | qp | probs | pbar | ppf
    float* qp = qs + H;
    float* pr = qp + 5 * NH;
    float* pb = pr + 64;
    float* pf = pb + 4 * NH;

    // ---- round trip 1: everything addressed by the node id
    const int h = lane / K, k = lane % K;
    const bool act = valid && lane < NH * K;
    const int g = a.group_idx[(size_t)node * K + k];
    const float* qrow = a.q + (size_t)node * a.ldq;
    for (int i = lane; i < H + 5 * NH; i += 64) qs[i] = qrow[i];
    if (lane < K) {
        const float4 f = reinterpret_cast<const float4*>(a.ppf)[(size_t)node * K + lane];
        pf[lane * 4 + 0] = f.x; pf[lane * 4 + 1] = f.y; pf[lane * 4 + 2] = f.z; pf[lane * 4 + 3] = f.w;
    }
    // ---- round trip 2: value rows (lane = channel) and key slices (lane = head, neighbour), all in flight together
    float vr[K][HV];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
        const int gk = __builtin_amdgcn_readlane(g, kk);  // lane kk (head 0) holds neighbour kk
        const float* vrow = a.v + (size_t)gk * a.ldv + lane;
#pragma unroll
        for (int i = 0; i < HV; ++i) vr[kk][i] = vrow[64 * i];
    }
    const float* krow = a.k + (size_t)g * a.ldk + (act ? h : 0) * c;
    __syncthreads();
    float score = -INFINITY;
    if (act) {
        const float* qh = qs + h * c;
        float dot = 0.f;
        for (int i = 0; i < c; i += 4) {
            const float4 kv = *reinterpret_cast<const float4*>(krow + i);
            const float4 qv = *reinterpret_cast<const float4*>(qh + i);
            dot += kv.x * qv.x; dot += kv.y * qv.y; dot += kv.z * qv.z; dot += kv.w * qv.w;
        }
        const float* qph = qp + h * 5;
        const float sp = qph[0] * pf[k * 4] + qph[1] * pf[k * 4 + 1] + qph[2] * pf[k * 4 + 2] + qph[3] * pf[k * 4 + 3] + qph[4];
        score = (dot + sp) * a.scale;
    }
    // softmax over k inside each K-lane group
    float mx = score;
#pragma unroll
    for (int o = 1; o < K; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float e = act ? expf(score - mx) : 0.f;  // accurate exp: the reference softmax is libm-exact
    float sum = e;
#pragma unroll
    for (int o = 1; o < K; o <<= 1) sum += __shfl_xor(sum, o, 64);
    const float p = act ? e / sum : 0.f;
    // pbar[h][j] = sum_k p * ppf[k][j]
    float b0 = p * pf[k * 4 + 0], b1 = p * pf[k * 4 + 1], b2 = p * pf[k * 4 + 2], b3 = p * pf[k * 4 + 3];
#pragma unroll
    for (int o = 1; o < K; o <<= 1) {
        b0 += __shfl_xor(b0, o, 64); b1 += __shfl_xor(b1, o, 64); b2 += __shfl_xor(b2, o, 64); b3 += __shfl_xor(b3, o, 64);
    }
    if (act) {
        pr[lane] = p;
        if (k == 0) { pb[h * 4 + 0] = b0; pb[h * 4 + 1] = b1; pb[h * 4 + 2] = b2; pb[h * 4 + 3] = b3; }
    }
    __syncthreads();
    if (!valid) return;

    // ---- phase 2: lane = channel, value rows already in registers
#pragma unroll
    for (int i = 0; i < HV; ++i) {
        const int ch = lane + 64 * i;
        const int hh = ch / c;
        float acc = 0.f;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) acc += pr[hh * K + kk] * vr[kk][i];
        const float4 w = reinterpret_cast<const float4*>(a.wvpe)[ch];
        acc += w.x * pb[hh * 4] + w.y * pb[hh * 4 + 1] + w.z * pb[hh * 4 + 2] + w.w * pb[hh * 4 + 3] + a.bvpe[ch];
        a.out[(size_t)node * a.ldo + ch] = acc;
    }
}

// Pfold (5*NH x H): row h*5+j holds Wpe[h*c + cc][j] (j<4) / bpe[h*c+cc] (j=4) at column h*c+cc, else 0.
// q_ext weights = [Wq ; Pfold @ Wq], bias = [bq ; Pfold @ bq]  (see header comment).
__global__ void build_pfold_kernel(int H, int NH, const float* __restrict__ wpe, const float* __restrict__ bpe, float* __restrict__ pf)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 5 * NH * H) return;
    const int row = t / H, col = t % H;
    const int h = row / 5, j = row % 5, c = H / NH;
    float v = 0.f;
    if (col / c == h) v = j < 4 ? wpe[col * 4 + j] : bpe[col];
    pf[t] = v;
}

}  // namespace

extern "C" int roitr_local_attention(const RoitrLocalAttn* a, hipStream_t stream)
{
    if (a->M <= 0) return ROITR_OK;
    const int c = a->H / a->heads;
    if (a->heads * a->K > 64 || (a->K & (a->K - 1)) || a->H % a->heads || c % 4 || (a->ldk % 4) || a->K < 1) return ROITR_ERR_UNSUPPORTED;
    const size_t per_wave = (size_t)a->H + 5 * a->heads + 64 + 4 * a->heads + 64;
    const size_t lds = per_wave * 4 * sizeof(float);
    if (per_wave % 4) return ROITR_ERR_UNSUPPORTED;  // keeps every wave's q row 16-byte aligned
    // algorithmic bytes: q row + K gathered k and v rows + ppf + idx in, one row out
    roitr_prof_begin(ROITR_PROF_LOCAL_ATTN, (double)a->M * ((a->H + 20.0) * 4 + a->K * (2.0 * a->H * 4 + 20.0) + a->H * 4.0), stream);
    const int hv = a->H / 64;
    if (a->H % 64 || (a->K != 8 && a->K != 16) || (hv != 1 && hv != 2 && hv != 4 && hv != 8)) return ROITR_ERR_UNSUPPORTED;
#define LA_CASE(KK, HH) local_attn_kernel<KK, HH><<<div_up(a->M, 4), 256, lds, stream>>>(*a)
    if (a->K == 8) { if (hv == 1) LA_CASE(8, 1); else if (hv == 2) LA_CASE(8, 2); else if (hv == 4) LA_CASE(8, 4); else LA_CASE(8, 8); }
    else { if (hv == 1) LA_CASE(16, 1); else if (hv == 2) LA_CASE(16, 2); else if (hv == 4) LA_CASE(16, 4); else LA_CASE(16, 8); }
#undef LA_CASE
    roitr_prof_end(ROITR_PROF_LOCAL_ATTN, stream);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_build_pfold(int H, int heads, const float* wpe, const float* bpe, float* pfold, hipStream_t stream)
{
    build_pfold_kernel<<<div_up(5L * heads * H, 256), 256, 0, stream>>>(H, heads, wpe, bpe, pfold);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}
