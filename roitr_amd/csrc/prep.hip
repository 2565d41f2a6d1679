// Input preparation in front of the model: per-point surface normals + orientation.
//
// Reference call sites: dataset/tdmatch.py:120-127, dataset/fdmatch.py:83-90, dataset/common.py:336-339
//     pcd.estimate_normals(search_param=o3d.geometry.KDTreeSearchParamKNN(knn=33))       (Open3D 0.13.0, requirements.txt:64)
//     normals = normal_redirect(points, normals, view_point)                             (dataset/common.py:312-320)
// Open3D is a third-party dependency that is not vendored in the reference tree; its published algorithm is restated
// here: for every point, the 33 nearest points (the point itself included), their 3x3 covariance from the raw
// cumulants in double precision (E[xx^T] - E[x]E[x]^T), and the unit eigenvector of the smallest eigenvalue from the
// closed-form symmetric 3x3 eigen-solve (trigonometric roots on the matrix scaled by its largest entry, eigenvector =
// the largest cross product of two rows of A - lambda I).  The sign Open3D leaves on that eigenvector is arbitrary;
// normal_redirect removes it (flip when (view_point - p) . n < 0), so the oriented normal is what parity is defined on.
//
// One lane per point; the kNN(33) comes from the grid / lane-per-query kernels of pointops_knn.hip (same exactness and
// tie rules as the model's grouping).
#include "common.h"
#include "roitr_pointops.h"

namespace {

__device__ __forceinline__ void cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// idx / dist2: (n, k) from the kNN; entries with dist2 >= 1e9 are the kNN's fill (cloud smaller than k)
__global__ __launch_bounds__(256) void normals_kernel(int n, int k, const float* __restrict__ xyz, const int* __restrict__ idx,
                                                      const float* __restrict__ dist2, float vx, float vy, float vz, int redirect,
                                                      float* __restrict__ normals)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int cnt = 0;
    for (int j = 0; j < k; ++j) {
        if (!(dist2[(size_t)p * k + j] < 1e9f)) continue;
        const int q = idx[(size_t)p * k + j];
        const double x = xyz[(size_t)q * 3], y = xyz[(size_t)q * 3 + 1], z = xyz[(size_t)q * 3 + 2];
        c[0] += x; c[1] += y; c[2] += z;
        c[3] += x * x; c[4] += x * y; c[5] += x * z; c[6] += y * y; c[7] += y * z; c[8] += z * z;
        ++cnt;
    }
    double nrm[3] = {0.0, 0.0, 1.0};   // Open3D's answer for degenerate neighbourhoods
    if (cnt >= 3) {
        const double inv = 1.0 / (double)cnt;
#pragma unroll
        for (int i = 0; i < 9; ++i) c[i] *= inv;
        double a00 = c[3] - c[0] * c[0], a11 = c[6] - c[1] * c[1], a22 = c[8] - c[2] * c[2];
        double a01 = c[4] - c[0] * c[1], a02 = c[5] - c[0] * c[2], a12 = c[7] - c[1] * c[2];
        const double mx = fmax(fmax(fmax(a00, a11), fmax(a22, a01)), fmax(a02, a12));
        if (mx > 0.0) {
            const double s = 1.0 / mx;
            a00 *= s; a11 *= s; a22 *= s; a01 *= s; a02 *= s; a12 *= s;
            const double off = a01 * a01 + a02 * a02 + a12 * a12;
            if (off > 0.0) {
                const double q = (a00 + a11 + a22) / 3.0;
                const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
                const double pp = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * off) / 6.0);
                const double c00 = b11 * b22 - a12 * a12, c01 = a01 * b22 - a12 * a02, c02 = a01 * a12 - b11 * a02;
                double hd = 0.5 * (b00 * c00 - a01 * c01 + a02 * c02) / (pp * pp * pp);
                hd = fmin(fmax(hd, -1.0), 1.0);
                const double ang = acos(hd) / 3.0;
                const double lmin = q + pp * 2.0 * cos(ang + 2.09439510239319549);   // the smallest of the three roots
                const double r0[3] = {a00 - lmin, a01, a02}, r1[3] = {a01, a11 - lmin, a12}, r2[3] = {a02, a12, a22 - lmin};
                double x01[3], x02[3], x12[3];
                cross3(r0, r1, x01); cross3(r0, r2, x02); cross3(r1, r2, x12);
                const double d01 = x01[0] * x01[0] + x01[1] * x01[1] + x01[2] * x01[2];
                const double d02 = x02[0] * x02[0] + x02[1] * x02[1] + x02[2] * x02[2];
                const double d12 = x12[0] * x12[0] + x12[1] * x12[1] + x12[2] * x12[2];
                const double* best = x01; double db = d01;
                if (d02 > db) { best = x02; db = d02; }
                if (d12 > db) { best = x12; db = d12; }
                if (db > 0.0) {
                    const double r = 1.0 / sqrt(db);
                    nrm[0] = best[0] * r; nrm[1] = best[1] * r; nrm[2] = best[2] * r;
                }
            } else {   // already diagonal: the axis of the smallest entry
                if (a00 < a11 && a00 < a22) { nrm[0] = 1.0; nrm[2] = 0.0; }
                else if (a11 < a00 && a11 < a22) { nrm[1] = 1.0; nrm[2] = 0.0; }
            }
        }
    }
    float nx = (float)nrm[0], ny = (float)nrm[1], nz = (float)nrm[2];
    if (redirect) {   // dataset/common.py:312-320, in the reference's float64 arithmetic
        const double px = xyz[(size_t)p * 3], py = xyz[(size_t)p * 3 + 1], pz = xyz[(size_t)p * 3 + 2];
        const double dotv = ((double)vx - px) * nrm[0] + ((double)vy - py) * nrm[1] + ((double)vz - pz) * nrm[2];
        if (dotv < 0.0) { nx = -nx; ny = -ny; nz = -nz; }
    }
    normals[(size_t)p * 3] = nx; normals[(size_t)p * 3 + 1] = ny; normals[(size_t)p * 3 + 2] = nz;
}

// dataset/common.py:312-320 on its own (float32 points / normals in, float64 dot like numpy's promotion there)
__global__ void redirect_kernel(int n, const float* __restrict__ xyz, const float* __restrict__ nin, float vx, float vy, float vz,
                                float* __restrict__ nout)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float a = nin[(size_t)p * 3], b = nin[(size_t)p * 3 + 1], c = nin[(size_t)p * 3 + 2];
    const double d = ((double)vx - xyz[(size_t)p * 3]) * a + ((double)vy - xyz[(size_t)p * 3 + 1]) * b + ((double)vz - xyz[(size_t)p * 3 + 2]) * c;
    const float s = d < 0.0 ? -1.f : 1.f;
    nout[(size_t)p * 3] = s * a; nout[(size_t)p * 3 + 1] = s * b; nout[(size_t)p * 3 + 2] = s * c;
}

}  // namespace

extern "C" size_t roitr_normals_workspace_bytes(int b, int n, int knn)
{
    // kNN workspace + idx (n, knn) + dist2 (n, knn)
    return roitr_knn_workspace_bytes(b, n, n) + (size_t)n * knn * 8 + 512;
}

extern "C" int roitr_estimate_normals(int b, int n, const float* xyz, const int* offset, int knn, int use_grid, const float* view_point,
                                      float* normals, void* ws, hipStream_t stream)
{
    if (n <= 0) return ROITR_OK;
    if (knn < 3 || knn > 100 || !xyz || !offset || !normals || !ws) return ROITR_ERR_ARG;
    const size_t kws = (roitr_knn_workspace_bytes(b, n, n) + 255) & ~(size_t)255;
    int* idx = reinterpret_cast<int*>(static_cast<char*>(ws) + kws);
    float* dist2 = reinterpret_cast<float*>(idx + (size_t)n * knn);
    int rc;
    if (use_grid) {
        rc = roitr_knn_build_grid_ex(b, n, n, xyz, offset, ws, (float)(knn + 1) / 3.0f, stream);
        if (rc != ROITR_OK) return rc;
    }
    rc = roitr_knnquery_ex(b, n, n, knn, xyz, xyz, offset, offset, idx, dist2, nullptr, nullptr, nullptr, nullptr, use_grid, n, ws, stream);
    if (rc != ROITR_OK) return rc;
    const float vx = view_point ? view_point[0] : 0.f, vy = view_point ? view_point[1] : 0.f, vz = view_point ? view_point[2] : 0.f;
    normals_kernel<<<div_up(n, 256), 256, 0, stream>>>(n, knn, xyz, idx, dist2, vx, vy, vz, view_point ? 1 : 0, normals);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}

extern "C" int roitr_normal_redirect(int n, const float* xyz, const float* normals_in, const float* view_point, float* normals_out,
                                     hipStream_t stream)
{
    if (n <= 0) return ROITR_OK;
    if (!view_point) return ROITR_ERR_ARG;
    redirect_kernel<<<div_up(n, 256), 256, 0, stream>>>(n, xyz, normals_in, view_point[0], view_point[1], view_point[2], normals_out);
    ROITR_LAUNCH_CHECK();
    return ROITR_OK;
}
