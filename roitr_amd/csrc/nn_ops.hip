// Row-wise HBM-bound layers of the RoITr path: residual + LayerNorm (+ReLU, +identity), L2 normalise,
// transpose (weight preparation), 3-NN interpolation, per-cloud mean, sinusoidal embedding, the
// max-over-k combine of the geometric structure embedding.  One wave per feature row where a row
// reduction is needed (C <= 1024: the row lives in registers, read once, written once), 16-byte
// accesses where the layout allows.
#include "common.h"
#include "roitr_engine.h"

namespace {

// ------------------------------------------------------------------ residual + LayerNorm
// out = act( LN(x + res[res_idx]) * gamma + beta  (+ post) )
// Reference call sites: attention.py:319 (norm(hidden + input[node_idx])), model/model.py:138-140
// (bn2 -> += identity -> relu), geoattention.py:50,161,241, nn.Sequential(Linear, LayerNorm, ReLU) of
// TransitionUp (model/model.py:89-97).  torch semantics: biased variance, eps inside the sqrt.
template <int VPL>  // values per lane: C <= 64*VPL
__global__ __launch_bounds__(256) void add_layernorm_kernel(int M, int C, const float* __restrict__ x, const float* __restrict__ res,
                                                            const int* __restrict__ res_idx, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ post, int relu,
                                                            float eps, float* __restrict__ out, const float* __restrict__ ip_feat = nullptr,
                                                            const int* __restrict__ ip_idx = nullptr, const float* __restrict__ ip_dist2 = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * C;
    const float* rr = nullptr;
    if (res) rr = res + (size_t)(res_idx ? res_idx[row] : row) * C;
    float v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        float t = 0.f;
        if (c < C) { t = xr[c]; if (rr) t += rr[c]; }
        v[i] = t; s += t;
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        const float d = c < C ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    // TransitionUp: + three-nearest-neighbour interpolation (expression order of interp3_add_kernel below), after the activation
    const float* f0 = nullptr; const float* f1 = nullptr; const float* f2 = nullptr;
    float w0 = 0.f, w1 = 0.f, w2 = 0.f;
    if (ip_feat) {
        const int* ii = ip_idx + (size_t)row * 3;
        const float* dd = ip_dist2 + (size_t)row * 3;
        w0 = 1.0f / (sqrtf(dd[0]) + 1e-8f); w1 = 1.0f / (sqrtf(dd[1]) + 1e-8f); w2 = 1.0f / (sqrtf(dd[2]) + 1e-8f);
        const float ws = (w0 + w1) + w2;
        w0 /= ws; w1 /= ws; w2 /= ws;
        f0 = ip_feat + (size_t)ii[0] * C; f1 = ip_feat + (size_t)ii[1] * C; f2 = ip_feat + (size_t)ii[2] * C;
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < C) {
            float y = (v[i] - mean) * rstd * gamma[c] + beta[c];
            if (post) y += post[(size_t)row * C + c];
            if (relu) y = fmaxf(y, 0.f);
            if (f0) {
                float acc = 0.f;
                acc += f0[c] * w0; acc += f1[c] * w1; acc += f2[c] * w2;
                y = y + acc;
            }
            out[(size_t)row * C + c] = y;
        }
    }
}

// F.normalize(x, p=2, dim=1) (RIGA_v2.py:64-65): x / max(||x||, 1e-12)
template <int VPL>
__global__ __launch_bounds__(256) void l2norm_kernel(int M, int C, const float* __restrict__ x, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < C ? x[(size_t)row * C + c] : 0.f;
        s += v[i] * v[i];
    }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < C) out[(size_t)row * C + c] = v[i] * inv;
    }
}

__global__ void transpose_kernel(int rows, int cols, const float* __restrict__ in, int ld_in, float* __restrict__ out, int ld_out)
{
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        tile[j][tx] = (r < rows && c < cols) ? in[(size_t)r * ld_in + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = bx + j, r = by + tx;
        if (c < cols && r < rows) out[(size_t)c * ld_out + r] = tile[tx][j];
    }
}

// pointops.interpolation (functions/pointops.py:168-182) + the `linear1(x1) +` of TransitionUp
// (model/model.py:116): out = base + ((f[i0]*w0 + f[i1]*w1) + f[i2]*w2), w = (1/(sqrt(d2)+1e-8)) normalised.
__global__ void interp3_add_kernel(long total, int C, const float* __restrict__ feat, const int* __restrict__ idx,
                                   const float* __restrict__ dist2, const float* __restrict__ base, float* __restrict__ out)
{
    for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += gridDim.x * 256L) {
        const long n = t / C;
        const int c = (int)(t % C);
        float w[3], s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) { w[i] = 1.0f / (sqrtf(dist2[n * 3 + i]) + 1e-8f); }
        s = (w[0] + w[1]) + w[2];
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) acc += feat[(size_t)idx[n * 3 + i] * C + c] * (w[i] / s);
        out[t] = (base ? base[t] : 0.f) + acc;
    }
}

// x_b.sum(0, True) / cnt per cloud (TransitionUp head, model/model.py:101-109)
__global__ __launch_bounds__(256) void segment_mean_kernel(int C, const float* __restrict__ x, const int* __restrict__ offset,
                                                           float* __restrict__ out)
{
    const int b = blockIdx.x;
    const int s = b == 0 ? 0 : offset[b - 1], e = offset[b];
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.f;
        for (int r = s; r < e; ++r) acc += x[(size_t)r * C + c];
        out[(size_t)b * C + c] = acc / (float)(e - s);
    }
}

// SinusoidalPositionalEmbedding (positional_encoding.py:38-62): out[r, 2t] = sin(v_r * div_t), out[r, 2t+1] = cos(.)
__global__ void sinusoid_kernel(long total_pairs, int half, const float* __restrict__ vals, const float* __restrict__ div_term,
                                float2* __restrict__ out)
{
    for (long t = blockIdx.x * 256L + threadIdx.x; t < total_pairs; t += gridDim.x * 256L) {
        const long r = t / half;
        const int i = (int)(t % half);
        const float om = vals[r] * div_term[i];
        out[t] = make_float2(sinf(om), cosf(om));
    }
}

// GeometricStructureEmbedding.forward tail (positional_encoding.py:146-152): E = P_d + max_k P_a[:, k, :]
__global__ void geo_combine_kernel(long total, int C, int k, const float* __restrict__ pd, const float* __restrict__ pa, float* __restrict__ out)
{
    for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += gridDim.x * 256L) {
        const long r = t / C;
        const int c = (int)(t % C);
        float m = pa[(r * k) * C + c];
        for (int j = 1; j < k; ++j) m = fmaxf(m, pa[(r * k + j) * C + c]);
        out[t] = pd[t] + m;
    }
}

__global__ void gather_rows_kernel(long total, int C, const float* __restrict__ in, const int* __restrict__ idx, int limit, float* __restrict__ out)
{
    for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += gridDim.x * 256L) {
        const long r = t / C;
        const int c = (int)(t % C);
        const int s = idx[r];
        out[t] = (s >= 0 && (limit <= 0 || s < limit)) ? in[(size_t)s * C + c] : 0.f;
    }
}

__global__ void compose_idx_kernel(int n, const int* __restrict__ outer, const int* __restrict__ inner, int* __restrict__ out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) out[t] = outer[inner[t]];
}

inline int ew_blocks(long total) { long b = (total + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int roitr_add_layernorm(int M, int C, const float* x, const float* res, const int* res_idx, const float* gamma,
                                   const float* beta, const float* post_add, int relu, float eps, float* out, hipStream_t stream)
{
    if (M <= 0) return ROITR_OK;
    if (C > 1024) return ROITR_ERR_UNSUPPORTED;
    const int blocks = div_up(M, 4);
#define LN_CASE(V) add_layernorm_kernel<V><<<blocks, 256, 0, stream>>>(M, C, x, res, res_idx, gamma, beta, post_add, relu, eps, out)
    if (C <= 64) LN_CASE(1); else if (C <= 128) LN_CASE(2); else if (C <= 256) LN_CASE(4); else if (C <= 512) LN_CASE(8); else LN_CASE(16);
#undef LN_CASE
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_add_layernorm_interp(int M, int C, const float* x, const float* res, const int* res_idx, const float* gamma,
                                          const float* beta, int relu, float eps, const float* ip_feat, const int* ip_idx,
                                          const float* ip_dist2, float* out, hipStream_t stream)
{
    if (M <= 0) return ROITR_OK;
    if (C > 1024 || !ip_feat || !ip_idx || !ip_dist2) return ROITR_ERR_UNSUPPORTED;
    const int blocks = div_up(M, 4);
#define LN_CASE(V) add_layernorm_kernel<V><<<blocks, 256, 0, stream>>>(M, C, x, res, res_idx, gamma, beta, nullptr, relu, eps, out, ip_feat, ip_idx, ip_dist2)
    if (C <= 64) LN_CASE(1); else if (C <= 128) LN_CASE(2); else if (C <= 256) LN_CASE(4); else if (C <= 512) LN_CASE(8); else LN_CASE(16);
#undef LN_CASE
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_l2_normalize(int M, int C, const float* x, float* out, hipStream_t stream)
{
    if (M <= 0) return ROITR_OK;
    if (C > 1024) return ROITR_ERR_UNSUPPORTED;
    const int blocks = div_up(M, 4);
    if (C <= 256) l2norm_kernel<4><<<blocks, 256, 0, stream>>>(M, C, x, out);
    else l2norm_kernel<16><<<blocks, 256, 0, stream>>>(M, C, x, out);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

namespace {
__global__ void add_vectors_kernel(int n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}
}  // namespace

/* out = a + b (n floats): bias vectors of folded layers */
extern "C" int roitr_add_vectors(int n, const float* a, const float* b, float* out, hipStream_t stream)
{
    if (n <= 0) return ROITR_OK;
    add_vectors_kernel<<<div_up(n, 256), 256, 0, stream>>>(n, a, b, out);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_transpose(int rows, int cols, const float* in, int ld_in, float* out, int ld_out, hipStream_t stream)
{
    if (rows <= 0 || cols <= 0) return ROITR_OK;
    transpose_kernel<<<dim3(div_up(cols, 32), div_up(rows, 32)), 256, 0, stream>>>(rows, cols, in, ld_in, out, ld_out);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_interp3_add(int n, int C, const float* feat, const int* idx, const float* dist2, const float* base, float* out,
                                 hipStream_t stream)
{
    const long total = (long)n * C;
    if (total <= 0) return ROITR_OK;
    interp3_add_kernel<<<ew_blocks(total), 256, 0, stream>>>(total, C, feat, idx, dist2, base, out);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_segment_mean(int b, int C, const float* x, const int* offset, float* out, hipStream_t stream)
{
    if (b <= 0) return ROITR_OK;
    segment_mean_kernel<<<b, 256, 0, stream>>>(C, x, offset, out);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_sinusoid(long rows, int C, const float* vals, const float* div_term, float* out, hipStream_t stream)
{
    const long total = rows * (C / 2);
    if (total <= 0) return ROITR_OK;
    sinusoid_kernel<<<ew_blocks(total), 256, 0, stream>>>(total, C / 2, vals, div_term, (float2*)out);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_geo_combine(long rows, int C, int k, const float* pd, const float* pa, float* out, hipStream_t stream)
{
    const long total = rows * C;
    if (total <= 0) return ROITR_OK;
    geo_combine_kernel<<<ew_blocks(total), 256, 0, stream>>>(total, C, k, pd, pa, out);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_gather_rows(long rows, int C, const float* in, const int* idx, int limit, float* out, hipStream_t stream)
{
    const long total = rows * C;
    if (total <= 0) return ROITR_OK;
    gather_rows_kernel<<<ew_blocks(total), 256, 0, stream>>>(total, C, in, idx, limit, out);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_compose_idx(int n, const int* outer, const int* inner, int* out, hipStream_t stream)
{
    if (n <= 0) return ROITR_OK;
    compose_idx_kernel<<<div_up(n, 256), 256, 0, stream>>>(n, outer, inner, out);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}
