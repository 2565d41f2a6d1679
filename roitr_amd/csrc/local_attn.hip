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
// One wave per node, 4 heads = the 4 DPP rows of the wave: lane l owns the HV = H/64 consecutive channels
// l*HV .. l*HV+HV-1, which all belong to head l/16.  Every gathered key / value row is read once, fully coalesced
// (HV floats per lane, scalar row base + lane offset: no per-lane address arithmetic); the per-head dot product is a
// per-lane partial + a 4-step DPP butterfly inside the 16-lane row; the PPF term rides in the same reduction (lanes
// 0..4 of each row add qp[h][t] * ppf[k][t] resp. the q_h . bpe_h constant); softmax is evaluated once per (head,
// neighbour) by lane t = neighbour of the head's row; the probabilities and the 4 pbar numbers per head cross lanes
// through a 320-byte LDS slot per wave.  No block barrier, no shuffles through LDS, gathered k/v rows are never
// materialised (the reference builds (M,K,H) copies with fancy indexing, attention.py:174-175).
#include "common.h"
#include "prof.h"
#include "roitr_engine.h"

namespace {

template <int HV> struct VecLoad;
template <> struct VecLoad<1> { static __device__ __forceinline__ void ld(const float* p, float* d) { d[0] = *p; }
                                static __device__ __forceinline__ void st(float* p, const float* d) { *p = d[0]; } };
template <> struct VecLoad<2> { static __device__ __forceinline__ void ld(const float* p, float* d) { const float2 t = *reinterpret_cast<const float2*>(p); d[0] = t.x; d[1] = t.y; }
                                static __device__ __forceinline__ void st(float* p, const float* d) { *reinterpret_cast<float2*>(p) = make_float2(d[0], d[1]); } };
template <> struct VecLoad<4> { static __device__ __forceinline__ void ld(const float* p, float* d) { const float4 t = *reinterpret_cast<const float4*>(p); d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
                                static __device__ __forceinline__ void st(float* p, const float* d) { *reinterpret_cast<float4*>(p) = make_float4(d[0], d[1], d[2], d[3]); } };
template <> struct VecLoad<8> { static __device__ __forceinline__ void ld(const float* p, float* d) { VecLoad<4>::ld(p, d); VecLoad<4>::ld(p + 4, d + 4); }
                                static __device__ __forceinline__ void st(float* p, const float* d) { VecLoad<4>::st(p, d); VecLoad<4>::st(p + 4, d + 4); } };

// bf16-stored rows (engine operand_dtype = bf16): HV consecutive bf16 -> fp32 (exact widening), fp32 -> bf16 (RNE) on store
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float x, float y)
{
    f32x2_t v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void unpack2(unsigned u, float& a, float& b) { a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xffff0000u); }
template <int HV> struct VecLoadH;
template <> struct VecLoadH<1> { static __device__ __forceinline__ void ld(const unsigned short* p, float* d) { d[0] = __uint_as_float((unsigned)*p << 16); }
                                 static __device__ __forceinline__ void st(unsigned short* p, const float* d) { *p = (unsigned short)(pack2(d[0], 0.f) & 0xffffu); } };
template <> struct VecLoadH<2> { static __device__ __forceinline__ void ld(const unsigned short* p, float* d) { unpack2(*reinterpret_cast<const unsigned*>(p), d[0], d[1]); }
                                 static __device__ __forceinline__ void st(unsigned short* p, const float* d) { *reinterpret_cast<unsigned*>(p) = pack2(d[0], d[1]); } };
template <> struct VecLoadH<4> { static __device__ __forceinline__ void ld(const unsigned short* p, float* d) { const uint2 t = *reinterpret_cast<const uint2*>(p); unpack2(t.x, d[0], d[1]); unpack2(t.y, d[2], d[3]); }
                                 static __device__ __forceinline__ void st(unsigned short* p, const float* d) { *reinterpret_cast<uint2*>(p) = make_uint2(pack2(d[0], d[1]), pack2(d[2], d[3])); } };
template <> struct VecLoadH<8> { static __device__ __forceinline__ void ld(const unsigned short* p, float* d) { VecLoadH<4>::ld(p, d); VecLoadH<4>::ld(p + 4, d + 4); }
                                 static __device__ __forceinline__ void st(unsigned short* p, const float* d) { VecLoadH<4>::st(p, d); VecLoadH<4>::st(p + 4, d + 4); } };
// row access in either storage type: `p` is the array base as the struct carries it (const float*), offsets in ELEMENTS
template <bool HALF, int HV> __device__ __forceinline__ void row_ld(const float* base, size_t off, float* d)
{
    if (HALF) VecLoadH<HV>::ld(reinterpret_cast<const unsigned short*>(base) + off, d);
    else VecLoad<HV>::ld(base + off, d);
}
template <bool HALF> __device__ __forceinline__ float elem_ld(const float* base, size_t off)
{
    if (HALF) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(base)[off] << 16);
    return base[off];
}

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// K = neighbours per node (8 / 16), HV = H / 64 (channels per lane); heads == 4.  HALF: q / k / v rows and the output row are
// stored in bf16 (RoitrLocalAttn::bf16); all arithmetic stays fp32.  NPW = nodes a wave works on at once: the kernel is bound
// by the latency of its two dependent gather round trips (ids -> rows), not by bytes (level 1: 10 TB/s out of L2 at full
// occupancy), so where the registers allow it (HV = 1) a wave keeps the loads of TWO nodes in flight; per node the arithmetic and
// its order are unchanged.
template <int K, int HV, bool HALF, int NPW>
__global__ __launch_bounds__(256) void local_attn_kernel(RoitrLocalAttn a)
{
    __shared__ __attribute__((aligned(16))) float xch[4][NPW][80];   // per wave and node: probs [head][16] | pbar [head][4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // nodes are visited in the grid's cell order when one is given, and every XCD gets a contiguous eighth of that
    // order: the k/v rows a node gathers are then re-used out of its XCD's L2 by the spatially adjacent nodes
    const int slot0 = (xcd_block_id((a.M + 4 * NPW - 1) / (4 * NPW)) * 4 + wave) * NPW;
    if (slot0 >= a.M) return;
    constexpr int H = 64 * HV;
    const int h = lane >> 4, t = lane & 15;
    int node[NPW]; bool live[NPW];
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        live[n] = slot0 + n < a.M;                       // wave-uniform
        const int sl = live[n] ? slot0 + n : slot0;      // a dead slot recomputes node slot0 and stores nothing
        int nd = sl;
        if (a.node_order) nd = __float_as_int(reinterpret_cast<const float4*>(a.node_order)[sl].w);
        node[n] = __builtin_amdgcn_readfirstlane(nd);
    }
    // ---- round trip 1: everything addressed by the node id
    int g[NPW]; float qv[NPW][HV], ec[NPW], pv[NPW][K];
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        g[n] = a.group_idx[(size_t)node[n] * K + (lane < K ? lane : 0)];
        const size_t qoff = (size_t)node[n] * a.ldq;
        row_ld<HALF, HV>(a.q, qoff + lane * HV, qv[n]);
        // qp[h][0..3] (PPF coefficients), qp[h][4] (q_h . bpe_h): extra columns of the q row, or formed below from wpe / bpe
        ec[n] = (!a.wpe && t < 5) ? elem_ld<HALF>(a.q, qoff + H + h * 5 + t) : 0.f;
        const float* pf = a.ppf + (size_t)node[n] * K * 4 + (t < 4 ? t : 0);
#pragma unroll
        for (int kk = 0; kk < K; ++kk) pv[n][kk] = pf[kk * 4];
    }
    // ---- round trip 2: key and value rows of the K neighbours of every node, all in flight together
    float kr[NPW][K][HV], vr[NPW][K][HV];
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int gk = __builtin_amdgcn_readlane(g[n], kk);
            row_ld<HALF, HV>(a.k, (size_t)gk * a.ldk + lane * HV, kr[n][kk]);
        }
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int gk = __builtin_amdgcn_readlane(g[n], kk);
            row_ld<HALF, HV>(a.v, (size_t)gk * a.ldv + lane * HV, vr[n][kk]);
        }
    }
    if (a.wpe) {   // kernel-argument uniform: qp[h] = [Wpe_h^T q_h, q_h . bpe_h] from this lane's channels + a row reduction
        float4 w4[HV]; float b1[HV];
#pragma unroll
        for (int i = 0; i < HV; ++i) { w4[i] = reinterpret_cast<const float4*>(a.wpe)[lane * HV + i]; b1[i] = a.bpe[lane * HV + i]; }
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f, p4 = 0.f;
#pragma unroll
            for (int i = 0; i < HV; ++i) {
                p0 = fmaf(w4[i].x, qv[n][i], p0); p1 = fmaf(w4[i].y, qv[n][i], p1); p2 = fmaf(w4[i].z, qv[n][i], p2);
                p3 = fmaf(w4[i].w, qv[n][i], p3); p4 = fmaf(b1[i], qv[n][i], p4);
            }
            p0 = row_allsum(p0); p1 = row_allsum(p1); p2 = row_allsum(p2); p3 = row_allsum(p3); p4 = row_allsum(p4);
            ec[n] = t == 0 ? p0 : (t == 1 ? p1 : (t == 2 ? p2 : (t == 3 ? p3 : (t == 4 ? p4 : 0.f))));
        }
    }
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        float* probs = xch[wave][n];
        float* pbar = xch[wave][n] + 64;
#pragma unroll
        for (int i = 0; i < HV; ++i) qv[n][i] *= a.scale;
        ec[n] *= a.scale;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) pv[n][kk] = t < 4 ? pv[n][kk] : (t == 4 ? 1.0f : 0.f);

        // ---- scores: s(h, kk) = scale * (q_h . k_h[kk] + qp_h . [ppf_kk, 1]); lane t of row h keeps s(h, t)
        float mine = 0.f, mx = -INFINITY;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            float d = ec[n] * pv[n][kk];
#pragma unroll
            for (int i = 0; i < HV; ++i) d = fmaf(qv[n][i], kr[n][kk][i], d);
            d = row_allsum(d);
            mine = t == kk ? d : mine;
            mx = fmaxf(mx, d);
        }
        const float e = t < K ? expf(mine - mx) : 0.f;   // accurate exp: the reference softmax is libm-exact
        const float p = e / row_allsum(e);
        probs[lane] = p;   // [h][t]
        lds_fence();
        float pk[K];
#pragma unroll
        for (int q4 = 0; q4 < K / 4; ++q4) {
            const float4 v4 = reinterpret_cast<const float4*>(probs + h * 16)[q4];
            pk[4 * q4] = v4.x; pk[4 * q4 + 1] = v4.y; pk[4 * q4 + 2] = v4.z; pk[4 * q4 + 3] = v4.w;
        }
        // pbar[h][j] = sum_k p(h,k) ppf[k][j]   (lanes t < 4 of every row)
        float pb = 0.f;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) pb = fmaf(pk[kk], pv[n][kk], pb);
        if (t < 4) pbar[h * 4 + t] = pb;
        lds_fence();
        const float4 pb4 = reinterpret_cast<const float4*>(pbar)[h];

        // ---- output: sum_k p v  +  Wvpe pbar + bvpe
        float o[HV], bias[HV];
        VecLoad<HV>::ld(a.bvpe + lane * HV, bias);
#pragma unroll
        for (int i = 0; i < HV; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int kk = 0; kk < K; ++kk) acc = fmaf(pk[kk], vr[n][kk][i], acc);
            const float4 w = reinterpret_cast<const float4*>(a.wvpe)[lane * HV + i];
            o[i] = acc + (w.x * pb4.x + w.y * pb4.y + w.z * pb4.z + w.w * pb4.w + bias[i]);
        }
        if (live[n]) {
            if (HALF) VecLoadH<HV>::st(reinterpret_cast<unsigned short*>(a.out) + (size_t)node[n] * a.ldo + lane * HV, o);
            else VecLoad<HV>::st(a.out + (size_t)node[n] * a.ldo + lane * HV, o);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// H = 64 form (level 1: 5.1 M nodes per call at 512 pairs).  In the kernel above a lane holds ONE channel at this width: every
// gathered row is a 256-byte wave instruction (60 vector-memory instructions for two nodes) and a wave keeps two nodes in flight.
// Here a node takes 16 lanes, a lane 4 consecutive channels (float4 accesses), a head = the 4 lanes of a DPP quad, and a wave works
// on FOUR nodes: the same bytes in a quarter of the memory instructions, twice the nodes in flight per wave, head reductions are
// two quad_perm adds instead of four row steps, and the softmax needs no LDS exchange (after the quad reduction every lane of a
// head holds all K scores).  fp32 rows, wpe / bpe given (the engine's default); per node the same formulas as above, summed in a
// different order.  Selected by shape only (H, K), never by M: a node's result does not depend on the batch it is in.
__device__ __forceinline__ float quad_allsum(float v)
{
    v = row_dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = row_dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    return v;
}
template <int CTRL> __device__ __forceinline__ float quad_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int K>
__global__ __launch_bounds__(256) void local_attn_quad_kernel(RoitrLocalAttn a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ns = lane >> 4, j = lane & 15, jq = j & 3;     // node slot of the wave, channel quad 4j..4j+3, position inside the head
    const int nblk = (a.M + 15) >> 4;
    const int slot = (xcd_block_id(nblk) * 4 + wave) * 4 + ns;
    if ((slot & ~3) >= a.M) return;                          // wave-uniform: all four slots of the wave are past the end
    const bool live = slot < a.M;
    const int sl = live ? slot : (slot & ~3);                // a dead slot recomputes the wave's first node and stores nothing
    int node = sl;
    if (a.node_order) node = __float_as_int(reinterpret_cast<const float4*>(a.node_order)[sl].w);
    // ---- round trip 1: addressed by the node id
    int g[K];
    {
        const int4* gp = reinterpret_cast<const int4*>(a.group_idx + (size_t)node * K);
#pragma unroll
        for (int q4 = 0; q4 < K / 4; ++q4) { const int4 t = gp[q4]; g[4 * q4] = t.x; g[4 * q4 + 1] = t.y; g[4 * q4 + 2] = t.z; g[4 * q4 + 3] = t.w; }
    }
    float4 qv = *reinterpret_cast<const float4*>(a.q + (size_t)node * a.ldq + 4 * j);
    float pv[K];
    {
        const float* pf = a.ppf + (size_t)node * K * 4 + jq;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) pv[kk] = pf[kk * 4];
    }
    // ---- round trip 2: key and value rows of the K neighbours, all in flight together
    float4 kr[K], vr[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) kr[kk] = *reinterpret_cast<const float4*>(a.k + (size_t)g[kk] * a.ldk + 4 * j);
#pragma unroll
    for (int kk = 0; kk < K; ++kk) vr[kk] = *reinterpret_cast<const float4*>(a.v + (size_t)g[kk] * a.ldv + 4 * j);
    __builtin_amdgcn_sched_barrier(0);   // the value rows are requested now, not sunk behind the softmax that does not need them
    // qp[h] = [Wpe_h^T q_h, q_h . bpe_h]: this lane's four channels, then the head's quad
    float ec, c4;
    {
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f, p4 = 0.f;
        const float qs[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 w = reinterpret_cast<const float4*>(a.wpe)[4 * j + i];
            const float b1 = a.bpe[4 * j + i];
            p0 = fmaf(w.x, qs[i], p0); p1 = fmaf(w.y, qs[i], p1); p2 = fmaf(w.z, qs[i], p2); p3 = fmaf(w.w, qs[i], p3); p4 = fmaf(b1, qs[i], p4);
        }
        p0 = quad_allsum(p0); p1 = quad_allsum(p1); p2 = quad_allsum(p2); p3 = quad_allsum(p3); p4 = quad_allsum(p4);
        ec = (jq == 0 ? p0 : (jq == 1 ? p1 : (jq == 2 ? p2 : p3))) * a.scale;
        c4 = jq == 0 ? p4 * a.scale : 0.f;                   // the q_h . bpe_h constant enters the reduction once per head
    }
    qv.x *= a.scale; qv.y *= a.scale; qv.z *= a.scale; qv.w *= a.scale;
    // ---- scores s(h, kk) = scale * (q_h . k_h[kk] + qp_h . [ppf_kk, 1]), softmax over kk in every lane of the head
    float sc[K];
    float mx = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
        float d = fmaf(ec, pv[kk], c4);
        d = fmaf(qv.x, kr[kk].x, d); d = fmaf(qv.y, kr[kk].y, d); d = fmaf(qv.z, kr[kk].z, d); d = fmaf(qv.w, kr[kk].w, d);
        d = quad_allsum(d);
        sc[kk] = d;
        mx = fmaxf(mx, d);
    }
    float sum = 0.f;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) { sc[kk] = expf(sc[kk] - mx); sum += sc[kk]; }   // accurate exp: the reference softmax is libm-exact
    // ---- pbar[h][t] = sum_k p(h,k) ppf[k][t] (lane jq = t), output = sum_k p v + Wvpe pbar + bvpe
    float pb = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
        const float p = sc[kk] / sum;
        pb = fmaf(p, pv[kk], pb);
        o.x = fmaf(p, vr[kk].x, o.x); o.y = fmaf(p, vr[kk].y, o.y); o.z = fmaf(p, vr[kk].z, o.z); o.w = fmaf(p, vr[kk].w, o.w);
    }
    const float b0 = quad_bcast<0x00>(pb), b1_ = quad_bcast<0x55>(pb), b2 = quad_bcast<0xAA>(pb), b3 = quad_bcast<0xFF>(pb);
    const float4 bias = *reinterpret_cast<const float4*>(a.bvpe + 4 * j);
    const float4 w0 = reinterpret_cast<const float4*>(a.wvpe)[4 * j], w1 = reinterpret_cast<const float4*>(a.wvpe)[4 * j + 1];
    const float4 w2 = reinterpret_cast<const float4*>(a.wvpe)[4 * j + 2], w3 = reinterpret_cast<const float4*>(a.wvpe)[4 * j + 3];
    o.x += w0.x * b0 + w0.y * b1_ + w0.z * b2 + w0.w * b3 + bias.x;
    o.y += w1.x * b0 + w1.y * b1_ + w1.z * b2 + w1.w * b3 + bias.y;
    o.z += w2.x * b0 + w2.y * b1_ + w2.z * b2 + w2.w * b3 + bias.z;
    o.w += w3.x * b0 + w3.y * b1_ + w3.z * b2 + w3.w * b3 + bias.w;
    if (live) *reinterpret_cast<float4*>(a.out + (size_t)node * a.ldo + 4 * j) = o;
}

// Pfold (5*NH x H): row h*5+j holds Wpe[h*c + cc][j] (j<4) / bpe[h*c+cc] (j=4) at column h*c+cc, else 0.
// q_ext weights = [Wq ; Pfold @ Wq], bias = [bq ; Pfold @ bq]  (see header comment).
// ---------------------------------------------------------------------------------------------------------------
// TransitionDown form with the key / value projections FOLDED INTO THE QUERY SIDE (round 3).  A TransitionDown transformer has
// M = N_in / 4 query nodes over the N_in points of the level above, 16 neighbours each (ppftransformer.py:227-253 behind
// model/model.py:56-80): the launch sequence above projects k | v for EVERY point -- a (N_in, 2H) tensor, 5.2 GB at level 2 of a
// 512-pair batch, written once and then gathered as 16 rows of 2H floats per node.  Both projections are linear in the point's
// input row x_j (k_j = Wk' x_j + bk', v_j = Wv' x_j + bv', Wk' / Wv' already folded with in_proj), so per head h
//     q_h . k_jh          = (Wk'_h^T q_h) . x_j + q_h . bk'_h       -- the second term does not depend on j: softmax drops it
//     sum_j a_hj v_jh     =  Wv'_h (sum_j a_hj x_j) + bv'_h          -- probabilities sum to 1
// Here: q~_h = Wk'_h^T q_h (I numbers per head and node, one batched GEMM in front), the kernel gathers the INPUT rows x_j (I floats
// instead of 2H = 4I), scores them against the four q~_h, and emits xbar_h = sum_j a_hj x_j (4 x I per node; one batched GEMM behind
// applies Wv'_h) next to the positional value term Wvpe_h pbar_h + bvpe_h.  Same FLOPs as the k | v GEMM it replaces (each point
// is gathered by 4 nodes on average), a quarter of the gather bytes, no (N_in, 2H) tensor.
// One wave per node, lane = V = I / 64 consecutive input channels.  The 16 neighbours x 4 heads = 64 wave-wide dot products of a
// node are reduced TOGETHER: v_permlane32_swap / v_permlane16_swap (gfx950) fold the four DPP rows while keeping one quarter of
// the values per row (row h ends with head h), row16_transpose_sum finishes inside the row -- lane (h, i) ends with the score of
// (head h, neighbour row16_slot(i)) in 63 adds instead of 64 x 6.  Softmax, pbar and the probability table are row-local after that.
__device__ __forceinline__ float row_allmax(float v)
{
    v = fmaxf(v, dpp_get<0xB1>(v)); v = fmaxf(v, dpp_get<0x4E>(v)); v = fmaxf(v, dpp_get<0x141>(v)); v = fmaxf(v, dpp_get<0x140>(v));
    return v;
}

template <int V, int HQ, int NPW>   // V = in_dim / 64 (x channels per lane), HQ = H / 64 (q / output channels per lane); K = 16, 4 heads
__global__ __launch_bounds__(256) void local_attn_fold_kernel(RoitrLocalAttnFold a)
{
    // NPW = nodes a wave works on at once (the gathers of all of them in flight together); per node the arithmetic and its order do
    // not depend on it.  The launcher uses NPW = 1 (see there).
    constexpr int K = 16, I = 64 * V, H = 64 * HQ;
    __shared__ __attribute__((aligned(16))) float probs[4][NPW][64];   // per wave and node: [head][neighbour]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot0 = (xcd_block_id((a.M + 4 * NPW - 1) / (4 * NPW)) * 4 + wave) * NPW;
    if (slot0 >= a.M) return;
    const int h = lane >> 4, i16 = lane & 15, kk_l = row16_slot(i16);
    int node[NPW]; bool live[NPW];
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        live[n] = slot0 + n < a.M;                       // wave-uniform
        const int sl = live[n] ? slot0 + n : slot0;      // a dead slot recomputes node slot0 and stores nothing
        int nd = sl;
        if (a.node_order) nd = __float_as_int(reinterpret_cast<const float4*>(a.node_order)[sl].w);
        node[n] = __builtin_amdgcn_readfirstlane(nd);
    }
    // ---- round trip 1: addressed by the node id
    int g[NPW]; float qt[NPW][4][V], qv[NPW][HQ]; float4 pf[NPW];
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        g[n] = a.group_idx[(size_t)node[n] * K + i16];
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) VecLoad<V>::ld(a.qt + (size_t)node[n] * (a.ldqt ? a.ldqt : 4 * I) + hh * I + lane * V, qt[n][hh]);
        VecLoad<HQ>::ld(a.q + (size_t)node[n] * a.ldq + lane * HQ, qv[n]);
        pf[n] = reinterpret_cast<const float4*>(a.ppf)[(size_t)node[n] * K + kk_l];   // this lane's neighbour
    }
    // ---- round trip 2: the input rows of the 16 neighbours of every node, all in flight together
    float xr[NPW][K][V];
#pragma unroll
    for (int n = 0; n < NPW; ++n)
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int gk = __builtin_amdgcn_readlane(g[n], kk);
            VecLoad<V>::ld(a.x + (size_t)gk * a.ldx + lane * V, xr[n][kk]);
        }
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        // u_h = Wpe_h^T q_h: the PPF coefficients of the score (q_h . bpe_h is constant over the neighbours: dropped with q_h . bk'_h)
        float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;
#pragma unroll
        for (int i = 0; i < HQ; ++i) {
            const float4 w = reinterpret_cast<const float4*>(a.wpe)[lane * HQ + i];
            u0 = fmaf(w.x, qv[n][i], u0); u1 = fmaf(w.y, qv[n][i], u1); u2 = fmaf(w.z, qv[n][i], u2); u3 = fmaf(w.w, qv[n][i], u3);
        }
        u0 = row_allsum(u0); u1 = row_allsum(u1); u2 = row_allsum(u2); u3 = row_allsum(u3);
        // ---- 64 partial dot products per lane, reduced over the wave in three transposing stages
        float z[16];
        {
            float w[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int h0 = i >> 4, k0 = i & 15;      // value i = (head h0, neighbour k0), value i + 32 = (head h0 + 2, neighbour k0)
                float d0 = 0.f, d1 = 0.f;
#pragma unroll
                for (int c = 0; c < V; ++c) { d0 = fmaf(qt[n][h0][c], xr[n][k0][c], d0); d1 = fmaf(qt[n][h0 + 2][c], xr[n][k0][c], d1); }
                w[i] = swap32_sum(d0, d1);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = swap16_sum(w[i], w[i + 16]);
        }
        // row h now holds, lane for lane, the 4-row sums of head h's 16 values
        const float tot = row16_transpose_sum(z, lane);
        const float s = (tot + (u0 * pf[n].x + u1 * pf[n].y + u2 * pf[n].z + u3 * pf[n].w)) * a.scale;
        const float mx = row_allmax(s);
        const float e = expf(s - mx);
        const float p = e / row_allsum(e);
        probs[wave][n][h * 16 + kk_l] = p;
        // pbar_h = sum_k p(h, k) ppf_k (every lane of row h)
        const float pb0 = row_allsum(p * pf[n].x), pb1 = row_allsum(p * pf[n].y), pb2 = row_allsum(p * pf[n].z), pb3 = row_allsum(p * pf[n].w);
        {
            float o[HQ], bias[HQ];
            VecLoad<HQ>::ld(a.bvpe + lane * HQ, bias);
#pragma unroll
            for (int i = 0; i < HQ; ++i) {
                const float4 w = reinterpret_cast<const float4*>(a.wvpe)[lane * HQ + i];
                o[i] = w.x * pb0 + w.y * pb1 + w.z * pb2 + w.w * pb3 + bias[i];
            }
            if (live[n]) VecLoad<HQ>::st(a.vpart + (size_t)node[n] * H + lane * HQ, o);
        }
        lds_fence();
        // ---- xbar_h = sum_k p(h, k) x_k: the probabilities of a head are wave-uniform LDS broadcasts
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            float acc[V];
#pragma unroll
            for (int c = 0; c < V; ++c) acc[c] = 0.f;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 p4 = reinterpret_cast<const float4*>(probs[wave][n])[hh * 4 + q4];
#pragma unroll
                for (int c = 0; c < V; ++c) {
                    acc[c] = fmaf(p4.x, xr[n][4 * q4][c], acc[c]); acc[c] = fmaf(p4.y, xr[n][4 * q4 + 1][c], acc[c]);
                    acc[c] = fmaf(p4.z, xr[n][4 * q4 + 2][c], acc[c]); acc[c] = fmaf(p4.w, xr[n][4 * q4 + 3][c], acc[c]);
                }
            }
            if (live[n]) VecLoad<V>::st(a.xbar + ((size_t)node[n] * 4 + hh) * I + lane * V, acc);
        }
    }
}

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
    const int hv = a->H / 64;
    // heads = the 4 DPP rows of a wave; float4 row accesses need 16-byte aligned rows
    if (a->heads != 4 || a->H % 64 || (a->K != 8 && a->K != 16) || (hv != 1 && hv != 2 && hv != 4 && hv != 8)) return ROITR_ERR_UNSUPPORTED;
    const bool half = a->bf16 != 0;
    if (half && a->bf16 != 3) return ROITR_ERR_UNSUPPORTED;   // bf16 rows in AND out, or fp32 both
    if (!half && hv > 1 && ((a->ldq | a->ldk | a->ldv | a->ldo) % (hv > 4 ? 4 : hv) || ((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->out) % 16))
        return ROITR_ERR_UNSUPPORTED;
    // bf16 rows: a lane reads HV bf16 = 2 HV bytes in pieces of at most 8 -> rows and bases aligned to min(2 HV, 8) bytes
    if (half && (((a->ldq | a->ldk | a->ldv | a->ldo) * 2) % (hv >= 4 ? 8 : 2 * hv) || ((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->out) % 8))
        return ROITR_ERR_UNSUPPORTED;
    // algorithmic bytes: q row + K gathered k and v rows + ppf + idx in, one row out
    roitr_prof_begin(ROITR_PROF_LOCAL_ATTN, (double)a->M * ((a->H + 20.0) * 4 + a->K * (2.0 * a->H * 4 + 20.0) + a->H * 4.0), stream);
#define LA_LAUNCH(KK, HH, NN)                                                                                             \
    do {                                                                                                                  \
        if (half) local_attn_kernel<KK, HH, true, NN><<<xcd_grid(div_up(a->M, 4 * NN)), 256, 0, stream>>>(*a);            \
        else local_attn_kernel<KK, HH, false, NN><<<xcd_grid(div_up(a->M, 4 * NN)), 256, 0, stream>>>(*a);                \
    } while (0)
    // two nodes per wave where a lane holds one channel (HV = 1: 64-wide levels), one otherwise (register budget)
#define LA_CASE(KK, HH)                                                                        \
    do {                                                                                       \
        if ((HH) == 1) LA_LAUNCH(KK, 1, 2);                                                     \
        else LA_LAUNCH(KK, HH, 1);                                                             \
    } while (0)
    // H = 64, fp32 rows, in-kernel qp: the 16-lanes-per-node form (round 3: the default for this shape; measured at the end of
    // round 2 at 512 pairs: the three level-1 launches 7.4 -> 6.5 ms).  The wave-per-node kernel serves every other shape.
    if (hv == 1 && !half && a->wpe && a->bpe && (a->K == 8 || a->K == 16) && ((a->ldq | a->ldk | a->ldv | a->ldo) & 3) == 0 &&
        (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->out | (uintptr_t)a->group_idx) & 15) == 0) {
        const int grid = xcd_grid(div_up(a->M, 16));
        if (a->K == 8) local_attn_quad_kernel<8><<<grid, 256, 0, stream>>>(*a);
        else local_attn_quad_kernel<16><<<grid, 256, 0, stream>>>(*a);
    } else
    if (a->K == 8) { if (hv == 1) LA_CASE(8, 1); else if (hv == 2) LA_CASE(8, 2); else if (hv == 4) LA_CASE(8, 4); else LA_CASE(8, 8); }
    else { if (hv == 1) LA_CASE(16, 1); else if (hv == 2) LA_CASE(16, 2); else if (hv == 4) LA_CASE(16, 4); else LA_CASE(16, 8); }
#undef LA_CASE
    roitr_prof_end(ROITR_PROF_LOCAL_ATTN, stream);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_local_attention_fold(const RoitrLocalAttnFold* a, hipStream_t stream)
{
    if (a->M <= 0) return ROITR_OK;
    const int v = a->in_dim / 64, hq = a->H / 64;
    if (a->in_dim % 64 || a->H % 64 || (v != 1 && v != 2 && v != 4) || (hq != 2 && hq != 4) || !a->wpe || !a->wvpe || !a->bvpe) {
        roitr_set_error("local_attention_fold: (in_dim, H) must be (64, 128), (128, 256) or (256, 256) with the folded PPF weights given", __FILE__, __LINE__);
        return ROITR_ERR_UNSUPPORTED;
    }
    if (a->ldqt != 0 && a->ldqt < 4 * a->in_dim) {   // ABI 2+ field: a client that does not zero the struct passes garbage here
        roitr_set_error("local_attention_fold: ldqt must be 0 (dense q~ rows) or >= 4 * in_dim", __FILE__, __LINE__);
        return ROITR_ERR_ARG;
    }
    if ((a->ldx | a->ldq | a->ldqt) % 4 || (((uintptr_t)a->x | (uintptr_t)a->q | (uintptr_t)a->qt | (uintptr_t)a->xbar | (uintptr_t)a->vpart | (uintptr_t)a->ppf |
                                   (uintptr_t)a->wpe | (uintptr_t)a->wvpe | (uintptr_t)a->bvpe) & 15)) {
        roitr_set_error("local_attention_fold: rows and arrays must be 16-byte aligned", __FILE__, __LINE__);
        return ROITR_ERR_UNSUPPORTED;
    }
    // algorithmic bytes: q and q~ rows, 16 gathered input rows, ppf + idx in; xbar and the positional value row out
    roitr_prof_begin(ROITR_PROF_LOCAL_ATTN, (double)a->M * (a->H * 4.0 + 4.0 * a->in_dim * 4 + 16.0 * (a->in_dim * 4.0 + 20.0) + 4.0 * a->in_dim * 4 + a->H * 4.0), stream);
    // one node per wave.  Measured at 512 pairs: two nodes per wave at V = 1 (NPW = 2: both nodes' gathers in flight together, 80 VGPRs)
    // 3.5 vs 3.0 ms for the level-2 launch beside the geometry stream, 89.7 vs 89.4 ms per step -- kept as a template parameter only
#define LF_CASE(VV, QQ, NN) local_attn_fold_kernel<VV, QQ, NN><<<xcd_grid(div_up(a->M, 4 * NN)), 256, 0, stream>>>(*a)
    if (v == 1 && hq == 2) LF_CASE(1, 2, 1);
    else if (v == 2 && hq == 4) LF_CASE(2, 4, 1);
    else if (v == 4 && hq == 4) LF_CASE(4, 4, 1);
    else { roitr_prof_end(ROITR_PROF_LOCAL_ATTN, stream); return ROITR_ERR_UNSUPPORTED; }
#undef LF_CASE
    roitr_prof_end(ROITR_PROF_LOCAL_ATTN, stream);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_local_attention_fold_supported(int in_dim, int H, int K)
{
    return K == 16 && ((in_dim == 64 && H == 128) || (in_dim == 128 && H == 256) || (in_dim == 256 && H == 256)) ? 1 : 0;
}

extern "C" int roitr_build_pfold(int H, int heads, const float* wpe, const float* bpe, float* pfold, hipStream_t stream)
{
    build_pfold_kernel<<<div_up(5L * heads * H, 256), 256, 0, stream>>>(H, heads, wpe, bpe, pfold);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}
