// GeometricStructureEmbedding.forward (reference model/transformer/positional_encoding.py:139-154) as a function table.
//
//     E[r, :] = proj_d(sinusoid(d_idx[r])) + max_k proj_a(sinusoid(a_idx[r, k]))
//
// proj_x(sinusoid(v)) is a Linear applied to [sin(v w_0), cos(v w_0), sin(v w_1), ...] (l.38-62) of ONE scalar v: each
// output channel is a univariate, band-limited function of v (largest angular frequency w_0 = 1),
//
//     g_c(v) = b_c + sum_i W[c, 2i] sin(v w_i) + W[c, 2i+1] cos(v w_i),
//
// so the (rows x C) x (C x C) GEMM per projection (2 (1 + k) C^2 flops per row, 29 ms per 512-pair step on the fp32 matrix
// cores) is not needed to evaluate it.  roitr_geo_table_build() fits, in float64 on the host and once per weight set, a
// degree-7 polynomial per channel on every interval [j h, (j + 1) h) of the value range (Chebyshev interpolation: the
// truncation term is (h/2)^8 / (2^7 8!) sum_i |coef_i| w_i^8 -- below fp32 rounding of g for h <= 2, and the builder MEASURES
// the fit error between the nodes and reports it); the kernel evaluates g_c(v) by Horner's rule in fp32 from an LDS-resident
// table.  Against the float64 function the result is closer than the fp32 GEMM form (1.2e-7 vs 1.7e-6 max abs at unit
// amplitude: no 256-term accumulation).  A value outside the tabulated range (a superpoint pair further apart than
// n_int_d * h * sigma_d) takes a direct sin / cos evaluation in the kernel, so the result is defined for any input.
//
// Layout: table[slice = C/64][interval (n_int_d distance intervals, then n_int_a angle intervals)][half: coefs 0..3 | 4..7][64
// channels][4 coefs].  A workgroup owns one 64-channel slice (its table slice stays in LDS: 2 KB per interval) and walks 64-row
// chunks; inside a chunk lane = channel, so the interval index / local coordinate of a row are wave-uniform (v_readlane), the 8
// coefficients of an evaluation are two conflict-free ds_read_b128 (round 5; before: [coef][channel] and four
// ds_read2st64_b32 at half the LDS rate) and the store of a row slice is one contiguous 256-byte line.  Bound: LDS bytes
// (4 evaluations x 8 coefficients per output) and the HBM write of E.
#include "common.h"
#include <mutex>
#include "prof.h"
#include <cmath>
#include <vector>

namespace {

constexpr int NCOEF = 8, CS = 64;

// same reduction as geo_embed.hip (branch-free Cody-Waite + minimax polynomials)
__device__ __forceinline__ void sincos_cw(float x, float& sn, float& cs)
{
    const float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(-k, 1.57079637050628662109375f, x);
    r = fmaf(-k, -4.37113900018624283e-8f, r);
    r = fmaf(-k, -1.71512449944201e-15f, r);
    const float z = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    ps = fmaf(ps * z, r, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    pc = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    const int q = (int)k;
    const float s1 = (q & 1) ? pc : ps, c1 = (q & 1) ? ps : pc;
    sn = (q & 2) ? -s1 : s1;
    cs = ((q + 1) & 2) ? -c1 : c1;
}

// out-of-table value: the projection row of this lane's channel against the sinusoid of x (k order, like the GEMM form)
__device__ __noinline__ float geo_direct(float x, const float* __restrict__ W, const float* __restrict__ bias,
                                         const float* __restrict__ div, int C, int col)
{
    const float* w = W + (size_t)col * C;
    float acc = 0.f;
    for (int i = 0; i < C / 2; ++i) {
        float sn, cs;
        sincos_cw(x * div[i], sn, cs);
        acc = fmaf(sn, w[2 * i], acc);
        acc = fmaf(cs, w[2 * i + 1], acc);
    }
    return acc + bias[col];
}

// p = the lane's float4 slot of the interval's low half (coefficients 0..3); the high half (4..7) lies CS float4 further on.
// Two ds_read_b128 per evaluation (consecutive lanes, consecutive 16-byte slots: conflict-free at 256 B per clock) instead of
// four ds_read2st64_b32 at 128 B per clock -- round 5: the kernel is bound by these reads.  Same Horner order: same bits.
__device__ __forceinline__ float poly8(const float4* __restrict__ p, float t)
{
    const float4 lo = p[0], hi = p[CS];
    float a = fmaf(hi.w, t, hi.z);
    a = fmaf(a, t, hi.y);
    a = fmaf(a, t, hi.x);
    a = fmaf(a, t, lo.w);
    a = fmaf(a, t, lo.z);
    a = fmaf(a, t, lo.y);
    return fmaf(a, t, lo.x);
}

// value -> (table interval or -1, local coordinate in [-1, 1))
__device__ __forceinline__ void locate(float x, float inv_h, int n, int base, int& j, float& t)
{
    const float u = x * inv_h;             // h is a power of two: exact
    const float fl = floorf(u);
    const bool ok = u >= 0.f && fl < (float)n;   // false for NaN
    j = ok ? base + (int)fl : -1;
    t = fmaf(2.f, u - fl, -1.f);           // u - floor(u) is exact
}

struct GeoTableArgs {
    long rows;
    int C, nd, na;
    float inv_h;
    const float *d_idx, *a_idx, *table, *div, *Wd, *bd, *Wa, *ba;
    void* out;
};

template <bool OH>
__device__ __forceinline__ void geo_table_body(const GeoTableArgs& g)
{
    extern __shared__ __attribute__((aligned(16))) float tab[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int slice = blockIdx.x, nint = g.nd + g.na;
    {
        const float4* src = reinterpret_cast<const float4*>(g.table + (size_t)slice * nint * (NCOEF * CS));
        float4* dst = reinterpret_cast<float4*>(tab);
        for (int i = tid; i < nint * (NCOEF * CS / 4); i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int col = slice * CS + lane;
    const float4* tl = reinterpret_cast<const float4*>(tab) + lane;
    const long nchunks = (g.rows + 63) >> 6;
    for (long q = (long)blockIdx.y * nwaves + wave; q < nchunks; q += (long)gridDim.y * nwaves) {
        const long row0 = q << 6, row = row0 + lane;
        float xd = 0.f, xa0 = 0.f, xa1 = 0.f, xa2 = 0.f;
        if (row < g.rows) {
            xd = g.d_idx[row];
            xa0 = g.a_idx[row * 3]; xa1 = g.a_idx[row * 3 + 1]; xa2 = g.a_idx[row * 3 + 2];
        }
        int jd, j0, j1, j2; float td, t0, t1, t2;
        locate(xd, g.inv_h, g.nd, 0, jd, td);
        locate(xa0, g.inv_h, g.na, g.nd, j0, t0);
        locate(xa1, g.inv_h, g.na, g.nd, j1, t1);
        locate(xa2, g.inv_h, g.na, g.nd, j2, t2);
        const int cnt = (int)min(64L, g.rows - row0);
#pragma unroll 2
        for (int e = 0; e < cnt; ++e) {
            const int sjd = __builtin_amdgcn_readlane(jd, e), sj0 = __builtin_amdgcn_readlane(j0, e);
            const int sj1 = __builtin_amdgcn_readlane(j1, e), sj2 = __builtin_amdgcn_readlane(j2, e);
            const float std_ = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(td), e));
            const float st0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(t0), e));
            const float st1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(t1), e));
            const float st2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(t2), e));
            float vd, v0, v1, v2;
            if (__builtin_expect((sjd | sj0 | sj1 | sj2) >= 0, 1)) {
                vd = poly8(tl + sjd * (2 * CS), std_);
                v0 = poly8(tl + sj0 * (2 * CS), st0);
                v1 = poly8(tl + sj1 * (2 * CS), st1);
                v2 = poly8(tl + sj2 * (2 * CS), st2);
            } else {   // some value of this row lies outside its table (wave-uniform branch)
                const float sxd = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(xd), e));
                const float sx0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(xa0), e));
                const float sx1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(xa1), e));
                const float sx2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(xa2), e));
                vd = sjd >= 0 ? poly8(tl + sjd * (2 * CS), std_) : geo_direct(sxd, g.Wd, g.bd, g.div, g.C, col);
                v0 = sj0 >= 0 ? poly8(tl + sj0 * (2 * CS), st0) : geo_direct(sx0, g.Wa, g.ba, g.div, g.C, col);
                v1 = sj1 >= 0 ? poly8(tl + sj1 * (2 * CS), st1) : geo_direct(sx1, g.Wa, g.ba, g.div, g.C, col);
                v2 = sj2 >= 0 ? poly8(tl + sj2 * (2 * CS), st2) : geo_direct(sx2, g.Wa, g.ba, g.div, g.C, col);
            }
            const float val = vd + fmaxf(v0, fmaxf(v1, v2));
            const size_t o = (size_t)(row0 + e) * g.C + col;
            if (OH) {
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                // lane = channel: the even lanes store their own and their right neighbour's value as one 4-byte word (round 6: 2-byte
                // stores are ~6x slower per byte than 4-byte ones on this part, MI355X_MICROARCH.md)
                const float vo = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(val), 0xB1, 0xf, 0xf, true));   // lane ^ 1
                f32x2_t pr = {val, vo};
                if ((threadIdx.x & 1) == 0)
                    *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(g.out) + o) = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2_t));
            } else reinterpret_cast<float*>(g.out)[o] = val;
        }
    }
}

__global__ __launch_bounds__(1024) void geo_table_kernel(GeoTableArgs g) { geo_table_body<false>(g); }
__global__ __launch_bounds__(1024) void geo_table_bf16o_kernel(GeoTableArgs g) { geo_table_body<true>(g); }

// float64 value of every channel of one projection at x
void exact_rows(double x, const std::vector<double>& div, const float* W, const float* b, int C, std::vector<double>& emb, double* out)
{
    for (int i = 0; i < C / 2; ++i) { emb[2 * i] = sin(x * div[i]); emb[2 * i + 1] = cos(x * div[i]); }
    for (int c = 0; c < C; ++c) {
        const float* w = W + (size_t)c * C;
        double s = 0.0;
        for (int k = 0; k < C; ++k) s += (double)w[k] * emb[k];
        out[c] = s + (double)b[c];
    }
}

}  // namespace

/* Number of floats of the table for C channels and n_int_d + n_int_a intervals. */
extern "C" size_t roitr_geo_table_floats(int C, int n_int_d, int n_int_a)
{
    return (size_t)(C / CS) * (size_t)(n_int_d + n_int_a) * NCOEF * CS;
}

/* HOST: fits the table (see the header of this file).  All pointers are HOST memory: div_term (C/2), Wd / Wa (C, C) row-major
 * (out_features, in_features), bd / ba (C), table = roitr_geo_table_floats() floats.  interval = h (a power of two).
 * fit[0..5] = {max |poly - g| of the distance projection over a dense probe of every interval (float64, 64 points), max |g_d|, the same two for the angle projection, then the largest PER-CHANNEL relative error (error of a channel
 * over that channel's own amplitude) of the distance and of the angle projection -- what roitr_engine_finalize gates on}. */
extern "C" int roitr_geo_table_build(int C, const float* div_term, const float* Wd, const float* bd, const float* Wa, const float* ba,
                                     float interval, int n_int_d, int n_int_a, float* table, double* fit)
{
    if (C % CS || C <= 0 || n_int_d < 1 || n_int_a < 1 || !(interval > 0.f)) return ROITR_ERR_ARG;
    {
        int ex; if (frexpf(interval, &ex) != 0.5f) return ROITR_ERR_ARG;   // power of two
    }
    const double h = interval;
    std::vector<double> div(C / 2), emb(C);
    for (int i = 0; i < C / 2; ++i) div[i] = (double)div_term[i];
    // Chebyshev nodes of the first kind and the monomial coefficients of T_0 .. T_7
    double node[NCOEF], tm[NCOEF][NCOEF] = {};
    for (int k = 0; k < NCOEF; ++k) node[k] = cos(M_PI * (2 * k + 1) / (2.0 * NCOEF));
    tm[0][0] = 1.0; tm[1][1] = 1.0;
    for (int m = 2; m < NCOEF; ++m)
        for (int p = 0; p < NCOEF; ++p) tm[m][p] = (p > 0 ? 2.0 * tm[m - 1][p - 1] : 0.0) - tm[m - 2][p];
    // the fit is MEASURED, densely: NPROBE points per interval (end points included) against the float64 function, per channel
    constexpr int NPROBE = 64;
    std::vector<double> f((size_t)NCOEF * C), mono((size_t)NCOEF * C), ref(C);
    std::vector<double> err_c[2] = {std::vector<double>(C, 0.0), std::vector<double>(C, 0.0)};
    std::vector<double> amp_c[2] = {std::vector<double>(C, 0.0), std::vector<double>(C, 0.0)};
    const int nint = n_int_d + n_int_a;
    for (int q = 0; q < 6; ++q) fit[q] = 0.0;
    for (int J = 0; J < nint; ++J) {
        const bool dist = J < n_int_d;
        const float* W = dist ? Wd : Wa; const float* b = dist ? bd : ba;
        const double x0 = ((dist ? J : J - n_int_d) + 0.5) * h;
        for (int k = 0; k < NCOEF; ++k) exact_rows(x0 + 0.5 * h * node[k], div, W, b, C, emb, &f[(size_t)k * C]);
        for (int c = 0; c < C; ++c) {
            double a[NCOEF];
            for (int m = 0; m < NCOEF; ++m) {
                double s = 0.0;
                for (int k = 0; k < NCOEF; ++k) s += f[(size_t)k * C + c] * cos(M_PI * m * (2 * k + 1) / (2.0 * NCOEF));
                a[m] = s * (m == 0 ? 1.0 : 2.0) / NCOEF;
            }
            for (int p = 0; p < NCOEF; ++p) {
                double s = 0.0;
                for (int m = p; m < NCOEF; ++m) s += a[m] * tm[m][p];
                mono[(size_t)p * C + c] = s;
                table[((((size_t)(c / CS) * nint + J) * 2 + p / 4) * CS + (c % CS)) * 4 + (p % 4)] = (float)s;
            }
        }
        for (int pi = 0; pi < NPROBE; ++pi) {
            const double t = -1.0 + 2.0 * pi / (double)(NPROBE - 1);
            exact_rows(x0 + 0.5 * h * t, div, W, b, C, emb, ref.data());
            for (int c = 0; c < C; ++c) {
                // the TRUNCATION error of the degree-7 fit (float64 coefficients): the fp32 rounding of the stored coefficients and of
                // the Horner steps is the same 2^-24-class noise any fp32 evaluation carries (tests/test_geo_table_cpu.py bounds it)
                double v = mono[(size_t)(NCOEF - 1) * C + c];
                for (int p = NCOEF - 2; p >= 0; --p) v = v * t + mono[(size_t)p * C + c];
                double& e = fit[dist ? 0 : 2]; double& amp = fit[dist ? 1 : 3];
                const double d = fabs(v - ref[c]);
                e = fmax(e, d); amp = fmax(amp, fabs(ref[c]));
                err_c[dist ? 0 : 1][c] = fmax(err_c[dist ? 0 : 1][c], d);
                amp_c[dist ? 0 : 1][c] = fmax(amp_c[dist ? 0 : 1][c], fabs(ref[c]));
            }
        }
    }
    // per channel: error relative to the channel's own amplitude (a channel that is 2^-10 of the projection's largest or less is
    // measured against that floor: it cannot matter more than that to anything downstream)
    for (int q = 0; q < 2; ++q) {
        const double floor_ = fit[q ? 3 : 1] / 1024.0;
        for (int c = 0; c < C; ++c) {
            const double a_ = fmax(amp_c[q][c], floor_);
            if (a_ > 0.0) fit[4 + q] = fmax(fit[4 + q], err_c[q][c] / a_);
        }
    }
    return ROITR_OK;
}

/* DEVICE: E (rows, C) from the table (fp32, or bf16 when out_bf16).  table / interval / n_int_* as built above and copied to
 * the device; div_term, Wd, bd, Wa, ba (device) serve values outside the table.  angle_k must be 3 and C a multiple of 64. */
extern "C" int roitr_geo_embed_table(long rows, int C, int angle_k, const float* d_idx, const float* a_idx, const float* table,
                                     float interval, int n_int_d, int n_int_a, const float* div_term, const float* Wd, const float* bd,
                                     const float* Wa, const float* ba, void* out, int out_bf16, hipStream_t stream)
{
    if (rows <= 0) return ROITR_OK;
    if (C % CS || angle_k != 3 || n_int_d < 1 || n_int_a < 1) return ROITR_ERR_UNSUPPORTED;
    const size_t lds = (size_t)(n_int_d + n_int_a) * NCOEF * CS * sizeof(float);
    if (lds > 160 * 1024) return ROITR_ERR_UNSUPPORTED;
    {   // dynamic-LDS limit of the two kernels: what THIS table needs, per device, raised when a larger table comes along
        static std::mutex mu;
        static size_t granted[64] = {0};
        int dev = 0;
        ROITR_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || lds > granted[dev]) {
            ROITR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(geo_table_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ROITR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(geo_table_bf16o_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < 64) granted[dev] = lds;
        }
    }
    GeoTableArgs g;
    g.rows = rows; g.C = C; g.nd = n_int_d; g.na = n_int_a; g.inv_h = 1.0f / interval;
    g.d_idx = d_idx; g.a_idx = a_idx; g.table = table; g.div = div_term; g.Wd = Wd; g.bd = bd; g.Wa = Wa; g.ba = ba; g.out = out;
    const int slices = C / CS;
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    static const int cus = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();   // one part per node
    const int threads = per_cu == 2 ? 512 : 1024;   // 16 waves per CU either way
    const long nchunks = (rows + 63) / 64;
    long gy = ((long)cus * per_cu + slices - 1) / slices;
    gy = std::max(1L, std::min(gy, (nchunks + threads / 64 - 1) / (threads / 64)));
    // "bytes" of the class: the HBM bytes of the launch (E written once, the four index values of a row read once per slice)
    roitr_prof_begin(ROITR_PROF_GEO_TABLE, (double)rows * C * (out_bf16 ? 2.0 : 4.0) + (double)rows * 16.0 * slices, stream);
    roitr_prof_note(ROITR_PROF_GEO_ALGO, 2.0 * rows * (1.0 + angle_k) * (double)C * C);
    if (out_bf16) geo_table_bf16o_kernel<<<dim3(slices, (unsigned)gy), threads, lds, stream>>>(g);
    else geo_table_kernel<<<dim3(slices, (unsigned)gy), threads, lds, stream>>>(g);
    roitr_prof_end(ROITR_PROF_GEO_TABLE, stream);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}
