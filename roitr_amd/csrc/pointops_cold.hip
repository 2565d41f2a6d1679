// The pointops entry points RoITr exports but never calls on its inference path:
// grouping / interpolation / subtraction / aggregation, forward and backward.  Built complete so the
// pybind-level API (pointops_api.cpp:15-22) has no holes.  All are HBM-bound gathers/scatters:
// one thread per output element (scalar 4-B accesses, consecutive threads on consecutive channels, so a
// wave reads/writes contiguous runs), grid-stride with 64-bit indexing.  Cold path: not vectorised.
//
// Ownership/zeroing conventions follow the reference: `+=`-style outputs must arrive zeroed
// (functions/pointops.py:146,199); backward scatters use atomics like the reference kernels.
#include "common.h"

namespace {

constexpr int TPB = 256;
inline int blocks_for(long work) { long b = (work + TPB - 1) / TPB; return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b)); }

// grouping_cuda_kernel.cu:5-15   out[m,s,c] = in[idx[m,s], c]
__global__ void grouping_fwd(long total, int nsample, int c, const float* __restrict__ in, const int* __restrict__ idx, float* __restrict__ out)
{
    for (long t = blockIdx.x * (long)TPB + threadIdx.x; t < total; t += (long)gridDim.x * TPB) {
        const int ci = (int)(t % c);
        const long ms = t / c;
        out[t] = in[(long)idx[ms] * c + ci];
    }
    (void)nsample;
}

// grouping_cuda_kernel.cu:17-27
__global__ void grouping_bwd(long total, int c, const float* __restrict__ gout, const int* __restrict__ idx, float* __restrict__ gin)
{
    for (long t = blockIdx.x * (long)TPB + threadIdx.x; t < total; t += (long)gridDim.x * TPB) {
        const int ci = (int)(t % c);
        const long ms = t / c;
        atomicAdd(gin + (long)idx[ms] * c + ci, gout[t]);
    }
}

// interpolation_cuda_kernel.cu:5-19   out[n,c] += sum_k in[idx[n,k],c] * w[n,k]   (k ascending)
__global__ void interp_fwd(long total, int c, int k, const float* __restrict__ in, const int* __restrict__ idx,
                           const float* __restrict__ w, float* __restrict__ out)
{
    for (long t = blockIdx.x * (long)TPB + threadIdx.x; t < total; t += (long)gridDim.x * TPB) {
        const int ci = (int)(t % c);
        const long ni = t / c;
        float acc = out[t];
        for (int i = 0; i < k; ++i) acc += in[(long)idx[ni * k + i] * c + ci] * w[ni * k + i];
        out[t] = acc;
    }
}

// interpolation_cuda_kernel.cu:21-35
__global__ void interp_bwd(long total, int c, int k, const float* __restrict__ gout, const int* __restrict__ idx,
                           const float* __restrict__ w, float* __restrict__ gin)
{
    for (long t = blockIdx.x * (long)TPB + threadIdx.x; t < total; t += (long)gridDim.x * TPB) {
        const int ci = (int)(t % c);
        const long ni = t / c;
        for (int i = 0; i < k; ++i) atomicAdd(gin + (long)idx[ni * k + i] * c + ci, gout[t] * w[ni * k + i]);
    }
}

// subtraction_cuda_kernel.cu:5-17   out[n,s,c] = in1[n,c] - in2[idx[n,s],c]
__global__ void sub_fwd(long total, int nsample, int c, const float* __restrict__ in1, const float* __restrict__ in2,
                        const int* __restrict__ idx, float* __restrict__ out)
{
    for (long t = blockIdx.x * (long)TPB + threadIdx.x; t < total; t += (long)gridDim.x * TPB) {
        const int ci = (int)(t % c);
        const long ns = t / c;
        const long ni = ns / nsample;
        out[t] = in1[ni * c + ci] - in2[(long)idx[ns] * c + ci];
    }
}

// subtraction_cuda_kernel.cu:19-32
__global__ void sub_bwd(long total, int nsample, int c, const int* __restrict__ idx, const float* __restrict__ gout,
                        float* __restrict__ g1, float* __restrict__ g2)
{
    for (long t = blockIdx.x * (long)TPB + threadIdx.x; t < total; t += (long)gridDim.x * TPB) {
        const int ci = (int)(t % c);
        const long ns = t / c;
        const long ni = ns / nsample;
        atomicAdd(g1 + ni * c + ci, gout[t]);
        atomicAdd(g2 + (long)idx[ns] * c + ci, -gout[t]);
    }
}

// aggregation_cuda_kernel.cu:5-21   out[n,c] += sum_s (in[idx[n,s],c] + pos[n,s,c]) * w[n,s,c % w_c]
__global__ void agg_fwd(long total, int nsample, int c, int w_c, const float* __restrict__ in, const float* __restrict__ pos,
                        const float* __restrict__ w, const int* __restrict__ idx, float* __restrict__ out)
{
    for (long t = blockIdx.x * (long)TPB + threadIdx.x; t < total; t += (long)gridDim.x * TPB) {
        const int ci = (int)(t % c);
        const long ni = t / c;
        const int wi = ci % w_c;
        float acc = out[t];
        for (int s = 0; s < nsample; ++s) {
            const long is = ni * nsample + s;
            acc += (in[(long)idx[is] * c + ci] + pos[is * c + ci]) * w[is * w_c + wi];
        }
        out[t] = acc;
    }
}

// aggregation_cuda_kernel.cu:23-43
__global__ void agg_bwd(long total, int nsample, int c, int w_c, const float* __restrict__ in, const float* __restrict__ pos,
                        const float* __restrict__ w, const int* __restrict__ idx, const float* __restrict__ gout,
                        float* __restrict__ gin, float* __restrict__ gpos, float* __restrict__ gw)
{
    for (long t = blockIdx.x * (long)TPB + threadIdx.x; t < total; t += (long)gridDim.x * TPB) {
        const int ci = (int)(t % c);
        const long ni = t / c;
        const int wi = ci % w_c;
        for (int s = 0; s < nsample; ++s) {
            const long is = ni * nsample + s;
            const long ii = (long)idx[is] * c + ci;
            atomicAdd(gin + ii, gout[t] * w[is * w_c + wi]);
            gpos[is * c + ci] = gout[t] * w[is * w_c + wi];
            atomicAdd(gw + is * w_c + wi, gout[t] * (in[ii] + pos[is * c + ci]));
        }
    }
}

}  // namespace

#define COLD_LAUNCH(kern, total, ...)                                        \
    do {                                                                     \
        if ((total) > 0) {                                                   \
            kern<<<blocks_for(total), TPB, 0, stream>>>(total, __VA_ARGS__); \
            ROITR_LAUNCH_CHECK();                                            \
        }                                                                    \
        return ROITR_OK;                                                     \
    } while (0)

extern "C" int roitr_grouping_forward(int m, int nsample, int c, const float* in, const int* idx, float* out, hipStream_t stream)
{ COLD_LAUNCH(grouping_fwd, (long)m * nsample * c, nsample, c, in, idx, out); }
extern "C" int roitr_grouping_backward(int m, int nsample, int c, const float* gout, const int* idx, float* gin, hipStream_t stream)
{ COLD_LAUNCH(grouping_bwd, (long)m * nsample * c, c, gout, idx, gin); }
extern "C" int roitr_interpolation_forward(int n, int c, int k, const float* in, const int* idx, const float* w, float* out, hipStream_t stream)
{ COLD_LAUNCH(interp_fwd, (long)n * c, c, k, in, idx, w, out); }
extern "C" int roitr_interpolation_backward(int n, int c, int k, const float* gout, const int* idx, const float* w, float* gin, hipStream_t stream)
{ COLD_LAUNCH(interp_bwd, (long)n * c, c, k, gout, idx, w, gin); }
extern "C" int roitr_subtraction_forward(int n, int nsample, int c, const float* in1, const float* in2, const int* idx, float* out, hipStream_t stream)
{ COLD_LAUNCH(sub_fwd, (long)n * nsample * c, nsample, c, in1, in2, idx, out); }
extern "C" int roitr_subtraction_backward(int n, int nsample, int c, const int* idx, const float* gout, float* g1, float* g2, hipStream_t stream)
{ COLD_LAUNCH(sub_bwd, (long)n * nsample * c, nsample, c, idx, gout, g1, g2); }
extern "C" int roitr_aggregation_forward(int n, int nsample, int c, int w_c, const float* in, const float* pos, const float* w, const int* idx, float* out, hipStream_t stream)
{ COLD_LAUNCH(agg_fwd, (long)n * c, nsample, c, w_c, in, pos, w, idx, out); }
extern "C" int roitr_aggregation_backward(int n, int nsample, int c, int w_c, const float* in, const float* pos, const float* w, const int* idx, const float* gout, float* gin, float* gpos, float* gw, hipStream_t stream)
{ COLD_LAUNCH(agg_bwd, (long)n * c, nsample, c, w_c, in, pos, w, idx, gout, gin, gpos, gw); }

// Exact legacy names (each *_cuda_kernel.h extern "C" block): void, default stream.
extern "C" void grouping_forward_cuda_launcher(int m, int nsample, int c, const float* in, const int* idx, float* out)
{ (void)roitr_grouping_forward(m, nsample, c, in, idx, out, nullptr); }
extern "C" void grouping_backward_cuda_launcher(int m, int nsample, int c, const float* gout, const int* idx, float* gin)
{ (void)roitr_grouping_backward(m, nsample, c, gout, idx, gin, nullptr); }
extern "C" void interpolation_forward_cuda_launcher(int n, int c, int k, const float* in, const int* idx, const float* w, float* out)
{ (void)roitr_interpolation_forward(n, c, k, in, idx, w, out, nullptr); }
extern "C" void interpolation_backward_cuda_launcher(int n, int c, int k, const float* gout, const int* idx, const float* w, float* gin)
{ (void)roitr_interpolation_backward(n, c, k, gout, idx, w, gin, nullptr); }
extern "C" void subtraction_forward_cuda_launcher(int n, int nsample, int c, const float* in1, const float* in2, const int* idx, float* out)
{ (void)roitr_subtraction_forward(n, nsample, c, in1, in2, idx, out, nullptr); }
extern "C" void subtraction_backward_cuda_launcher(int n, int nsample, int c, const int* idx, const float* gout, float* g1, float* g2)
{ (void)roitr_subtraction_backward(n, nsample, c, idx, gout, g1, g2, nullptr); }
extern "C" void aggregation_forward_cuda_launcher(int n, int nsample, int c, int w_c, const float* in, const float* pos, const float* w, const int* idx, float* out)
{ (void)roitr_aggregation_forward(n, nsample, c, w_c, in, pos, w, idx, out, nullptr); }
extern "C" void aggregation_backward_cuda_launcher(int n, int nsample, int c, int w_c, const float* in, const float* pos, const float* w, const int* idx, const float* gout, float* gin, float* gpos, float* gw)
{ (void)roitr_aggregation_backward(n, nsample, c, w_c, in, pos, w, idx, gout, gin, gpos, gw, nullptr); }
