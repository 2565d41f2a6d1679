"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the input-preparation step.

normal_redirect follows dataset/common.py:312-320 line by line (pinned by tests/golden/prep_eval.npz, captured from the
imported reference).  estimate_normals restates the published algorithm of Open3D 0.13.0 (requirements.txt:64; a
third-party dependency that is NOT in the reference tree and not installed here: **parity unpinned** for that half):
PointCloud::EstimateNormals with KDTreeSearchParamKNN(knn): the knn nearest points (query point included), covariance
from the raw cumulants in float64, unit eigenvector of the smallest eigenvalue.  Open3D's sign of that vector is
arbitrary; callers compare after normal_redirect.
"""
import numpy as np


def normal_redirect(points, normals, view_point):
    vec_dot = np.sum((view_point - points) * normals, axis=-1)   # dataset/common.py:316
    mask = vec_dot < 0.0
    out = normals.copy()
    out[mask] *= -1.0
    return out


def knn_bruteforce(points, k):
    """indices (n, min(k, n)) of the nearest points, ascending distance (float64 distances of the float32 inputs)."""
    p = points.astype(np.float64)
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    return np.argsort(d2, axis=1, kind="stable")[:, :k]


def estimate_normals(points, knn=33):
    """(n,3) float64 unit normals and the eigenvalue gap ratio (lambda_1 - lambda_0) / lambda_2 used by the tests to skip
    points whose normal direction is ill-conditioned."""
    pts = points.astype(np.float64)
    nbr = knn_bruteforce(points, knn)
    n = pts.shape[0]
    normals = np.zeros((n, 3))
    gap = np.zeros(n)
    for i in range(n):
        q = pts[nbr[i]]
        if q.shape[0] < 3:
            normals[i] = (0.0, 0.0, 1.0)
            continue
        mean = q.mean(0)
        cov = (q[:, :, None] * q[:, None, :]).mean(0) - np.outer(mean, mean)   # cumulant form, like Open3D
        w, v = np.linalg.eigh(cov)
        normals[i] = v[:, 0]
        gap[i] = (w[1] - w[0]) / max(w[2], 1e-300)
    return normals, gap


def estimate_normals_fast(points, knn=33, view_point=(0.0, 0.0, 0.0), threads=1):
    """Vectorised form for timing / larger clouds: kNN from the C restatement (oracle/pointops_ref.c), batched eigh."""
    from . import pointops_cpu as OP
    n = points.shape[0]
    off = np.array([n], dtype=np.int32)
    idx, _ = OP.knnquery_raw(min(knn, n), points, points, off, off, threads=threads)
    q = points.astype(np.float64)[idx.astype(np.int64)]                       # (n, k, 3)
    mean = q.mean(1)
    cov = np.einsum("nki,nkj->nij", q, q) / q.shape[1] - mean[:, :, None] * mean[:, None, :]
    w, v = np.linalg.eigh(cov)
    nrm = v[:, :, 0]
    return normal_redirect(points.astype(np.float64), nrm, np.asarray(view_point, dtype=np.float64))
