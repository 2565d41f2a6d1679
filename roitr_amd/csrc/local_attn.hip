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

__global__ __launch_bounds__(256) void local_attn_kernel(RoitrLocalAttn a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int node = blockIdx.x * 4 + wave;
    const int H = a.H, K = a.K, NH = a.heads, c = H / NH;
    float* qs = smem + (size_t)wave * (H + 5 * NH + 64 + 4 * NH + 64);  // q row | qp | probs | pbar | ppf
    float* qp = qs + H;
    float* pr = qp + 5 * NH;
    float* pb = pr + 64;
    float* pf = pb + 4 * NH;
    const bool valid = node < a.M;  // whole-wave predicate (no early exit: block barriers below)

    const float* qrow = a.q + (size_t)(valid ? node : 0) * a.ldq;
    for (int i = lane; i < H + 5 * NH; i += 64) qs[i] = valid ? qrow[i] : 0.f;
    if (lane < K && valid) {
        const float4 f = reinterpret_cast<const float4*>(a.ppf)[(size_t)node * K + lane];
        pf[lane * 4 + 0] = f.x; pf[lane * 4 + 1] = f.y; pf[lane * 4 + 2] = f.z; pf[lane * 4 + 3] = f.w;
    }
    __syncthreads();

    // ---- phase 1: lane = h*K + k
    const int h = lane / K, k = lane % K;
    const bool act = valid && lane < NH * K;
    float score = -INFINITY;
    int g = 0;
    if (act) {
        g = a.group_idx[(size_t)node * K + k];
        const float* krow = a.k + (size_t)g * a.ldk + h * c;
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
    // softmax over k inside each K-lane group (K is a power of two: 8 or 16)
    float mx = score;
    for (int o = 1; o < K; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float e = act ? expf(score - mx) : 0.f;  // accurate exp: the reference softmax is libm-exact
    float sum = e;
    for (int o = 1; o < K; o <<= 1) sum += __shfl_xor(sum, o, 64);
    const float p = act ? e / sum : 0.f;
    // pbar[h][j] = sum_k p * ppf[k][j]
    float b0 = p * pf[(act ? k : 0) * 4 + 0], b1 = p * pf[(act ? k : 0) * 4 + 1], b2 = p * pf[(act ? k : 0) * 4 + 2],
          b3 = p * pf[(act ? k : 0) * 4 + 3];
    for (int o = 1; o < K; o <<= 1) {
        b0 += __shfl_xor(b0, o, 64); b1 += __shfl_xor(b1, o, 64); b2 += __shfl_xor(b2, o, 64); b3 += __shfl_xor(b3, o, 64);
    }
    if (act) {
        pr[lane] = p;
        if (k == 0) { pb[h * 4 + 0] = b0; pb[h * 4 + 1] = b1; pb[h * 4 + 2] = b2; pb[h * 4 + 3] = b3; }
    }
    // neighbour indices for phase 2, one per lane k < K
    int* gi = reinterpret_cast<int*>(pf);  // ppf no longer needed after pbar: reuse
    __syncthreads();
    if (act && h == 0) gi[k] = g;
    __syncthreads();
    if (!valid) return;

    // ---- phase 2: lane = channel
    for (int ch = lane; ch < H; ch += 64) {
        const int hh = ch / c;
        float acc = 0.f;
        for (int kk = 0; kk < K; ++kk) acc += pr[hh * K + kk] * a.v[(size_t)gi[kk] * a.ldv + ch];
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
    local_attn_kernel<<<div_up(a->M, 4), 256, lds, stream>>>(*a);
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
