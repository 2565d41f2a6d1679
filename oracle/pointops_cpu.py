"""ctypes/numpy front-end to oracle/pointops_ref.c -- TEST INFRASTRUCTURE ONLY.

Restates the glue of cpp_wrappers/pointops/functions/pointops.py (reference) on numpy
arrays: output allocation, the 1e10 fill of `tmp` (pointops.py:22), n_max (pointops.py:18-20),
sqrt of the kNN distances (pointops.py:43), queryandgroup's "k+1 then drop column 0"
(pointops.py:88-89) and the Python `interpolation` (pointops.py:168-182).

Parity unpinned for the native half (see pointops_ref.c header).
"""
import ctypes
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_pointops.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "pointops_ref.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_opt_n_threads.restype = ctypes.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i)


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ci32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def opt_n_threads(n):
    return lib().oracle_opt_n_threads(int(n))


def furthestsampling(xyz, offset, new_offset):
    """pointops.py:10-27.  Returns int32 (m,)."""
    xyz, offset, new_offset = _c32(xyz), _ci32(offset), _ci32(new_offset)
    b = offset.shape[0]
    n_max = int(offset[0])
    for i in range(1, b):
        n_max = max(int(offset[i] - offset[i - 1]), n_max)
    idx = np.zeros(int(new_offset[b - 1]), dtype=np.int32)
    tmp = np.full(xyz.shape[0], 1e10, dtype=np.float32)
    lib().oracle_furthestsampling(b, n_max, _fp(xyz), _ip(offset), _ip(new_offset), _fp(tmp), _ip(idx))
    return idx


def knnquery_raw(nsample, xyz, new_xyz, offset, new_offset, threads=1):
    """The native call alone: int32 idx (m,k) and SQUARED fp32 distances (m,k)."""
    xyz = _c32(xyz)
    new_xyz = xyz if new_xyz is None else _c32(new_xyz)
    offset, new_offset = _ci32(offset), _ci32(new_offset)
    m = new_xyz.shape[0]
    assert nsample <= 100  # knnquery_cuda_kernel.cu:86-87
    idx = np.zeros((m, nsample), dtype=np.int32)
    d2 = np.zeros((m, nsample), dtype=np.float32)
    L = lib()
    if threads <= 1 or m < 4 * threads:
        L.oracle_knnquery(m, nsample, _fp(xyz), _fp(new_xyz), _ip(offset), _ip(new_offset), _ip(idx), _fp(d2))
    else:
        # ctypes releases the GIL: plain threads give real parallelism over query ranges
        bounds = np.linspace(0, m, threads + 1).astype(int)

        def work(a, b):
            L.oracle_knnquery_range(int(a), int(b), nsample, _fp(xyz), _fp(new_xyz), _ip(offset),
                                    _ip(new_offset), _ip(idx), _fp(d2))
        ts = [threading.Thread(target=work, args=(bounds[t], bounds[t + 1])) for t in range(threads)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    return idx, d2


def knnquery(nsample, xyz, new_xyz, offset, new_offset, threads=1):
    """pointops.py:30-45: (idx int32, euclidean distance fp32)."""
    idx, d2 = knnquery_raw(nsample, xyz, new_xyz, offset, new_offset, threads)
    return idx, np.sqrt(d2)


def queryandgroup_idx(nsample, xyz, new_xyz, offset, new_offset, threads=1):
    """pointops.py:79-92 with idx=None, return_idx=True: kNN(nsample+1), drop column 0, int64."""
    idx, _ = knnquery(nsample + 1, xyz, new_xyz, offset, new_offset, threads)
    return np.ascontiguousarray(idx[:, 1:]).astype(np.int64)


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """pointops.py:168-182."""
    idx, dist = knnquery(k, xyz, new_xyz, offset, new_offset)
    dist_recip = (np.float32(1.0) / (dist + np.float32(1e-8))).astype(np.float32)
    norm = dist_recip.sum(axis=1, keepdims=True, dtype=np.float32)
    weight = dist_recip / norm
    feat = _c32(feat)
    out = np.zeros((new_xyz.shape[0], feat.shape[1]), dtype=np.float32)
    for i in range(k):
        out += feat[idx[:, i].astype(np.int64), :] * weight[:, i:i + 1]
    return out


def grouping_forward(inp, idx):
    inp, idx = _c32(inp), _ci32(idx)
    m, ns = idx.shape
    c = inp.shape[1]
    out = np.empty((m, ns, c), dtype=np.float32)
    lib().oracle_grouping_forward(m, ns, c, _fp(inp), _ip(idx), _fp(out))
    return out


def grouping_backward(grad_out, idx, n):
    grad_out, idx = _c32(grad_out), _ci32(idx)
    m, ns, c = grad_out.shape
    gi = np.zeros((n, c), dtype=np.float32)
    lib().oracle_grouping_backward(m, ns, c, _fp(grad_out), _ip(idx), _fp(gi))
    return gi


def interpolation_forward(inp, idx, weight):
    inp, idx, weight = _c32(inp), _ci32(idx), _c32(weight)
    n, k = idx.shape
    c = inp.shape[1]
    out = np.zeros((n, c), dtype=np.float32)
    lib().oracle_interpolation_forward(n, c, k, _fp(inp), _ip(idx), _fp(weight), _fp(out))
    return out


def interpolation_backward(grad_out, idx, weight, m):
    grad_out, idx, weight = _c32(grad_out), _ci32(idx), _c32(weight)
    n, c = grad_out.shape
    k = idx.shape[1]
    gi = np.zeros((m, c), dtype=np.float32)
    lib().oracle_interpolation_backward(n, c, k, _fp(grad_out), _ip(idx), _fp(weight), _fp(gi))
    return gi


def subtraction_forward(in1, in2, idx):
    in1, in2, idx = _c32(in1), _c32(in2), _ci32(idx)
    n, c = in1.shape
    ns = idx.shape[1]
    out = np.zeros((n, ns, c), dtype=np.float32)
    lib().oracle_subtraction_forward(n, ns, c, _fp(in1), _fp(in2), _ip(idx), _fp(out))
    return out


def subtraction_backward(idx, grad_out, n2=None):
    idx, grad_out = _ci32(idx), _c32(grad_out)
    n, ns, c = grad_out.shape
    g1 = np.zeros((n, c), dtype=np.float32)
    g2 = np.zeros((n if n2 is None else n2, c), dtype=np.float32)
    lib().oracle_subtraction_backward(n, ns, c, _ip(idx), _fp(grad_out), _fp(g1), _fp(g2))
    return g1, g2


def aggregation_forward(inp, position, weight, idx):
    inp, position, weight, idx = _c32(inp), _c32(position), _c32(weight), _ci32(idx)
    n, ns, c = position.shape
    w_c = weight.shape[-1]
    out = np.zeros((n, c), dtype=np.float32)
    lib().oracle_aggregation_forward(n, ns, c, w_c, _fp(inp), _fp(position), _fp(weight), _ip(idx), _fp(out))
    return out


def aggregation_backward(inp, position, weight, idx, grad_out):
    inp, position, weight, idx, grad_out = _c32(inp), _c32(position), _c32(weight), _ci32(idx), _c32(grad_out)
    n, ns, c = position.shape
    w_c = weight.shape[-1]
    gi = np.zeros_like(inp)
    gp = np.zeros_like(position)
    gw = np.zeros_like(weight)
    lib().oracle_aggregation_backward(n, ns, c, w_c, _fp(inp), _fp(position), _fp(weight), _ip(idx),
                                      _fp(grad_out), _fp(gi), _fp(gp), _fp(gw))
    return gi, gp, gw
