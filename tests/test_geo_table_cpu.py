"""Host half of the function-table form of GeometricStructureEmbedding (csrc/geo_table.hip): the table builder is plain C++
inside libroitr_hip.so and runs without a GPU.  The polynomial evaluation of the kernel (fp32 Horner on the fp32 table) is
restated in numpy here and compared with a float64 evaluation of positional_encoding.py:38-62 + the two Linear layers."""
import numpy as np
import pytest
import torch

C = 256


def _weights(seed, scale):
    rng = np.random.default_rng(seed)
    div = np.exp(np.arange(0, C, 2, dtype=np.float32) * np.float32(-np.log(10000.0) / C)).astype(np.float32)
    wd, wa = (rng.normal(size=(C, C)) * scale).astype(np.float32), (rng.normal(size=(C, C)) * scale).astype(np.float32)
    bd, ba = rng.normal(size=C).astype(np.float32), rng.normal(size=C).astype(np.float32)
    return div, wd, bd, wa, ba


def _exact(x, div, w, b):
    om = x.astype(np.float32).astype(np.float64)[:, None] * div.astype(np.float64)[None]
    emb = np.empty((len(x), C))
    emb[:, 0::2], emb[:, 1::2] = np.sin(om), np.cos(om)
    return emb @ w.astype(np.float64).T + b


def _horner(table, n_int, x, h, base):
    T = table.reshape(C // 64, n_int, 2, 64, 4).transpose(0, 1, 2, 4, 3).reshape(C // 64, n_int, 8, 64)   # [half][channel][4] -> [coef][channel]
    x = x.astype(np.float32)
    u = x * np.float32(1.0 / h)
    fl = np.floor(u)
    t = (np.float32(2) * (u - fl) - np.float32(1)).astype(np.float32)
    co = T[:, base + fl.astype(int)]   # (slices, n, 8, 64)
    acc = co[:, :, 7]
    for p in range(6, -1, -1):
        acc = (acc * t[None, :, None] + co[:, :, p]).astype(np.float32)
    return acc.transpose(1, 0, 2).reshape(len(x), C)


@pytest.mark.parametrize("h", [2.0, 1.0])
@pytest.mark.parametrize("scale", [1.0 / 16, 0.4])
def test_table_fit_is_below_fp32_rounding(h, scale):
    from roitr_amd import ops
    div, wd, bd, wa, ba = _weights(7, scale)
    t = torch.from_numpy
    table, nd, na, fit = ops.geo_table_build(t(div), t(wd), t(bd), t(wa), t(ba), interval=h, d_range=48.0, a_range=12.0)
    assert nd == int(np.ceil(48.0 / h)) and na == int(12.0 // h) + 1
    assert table.numel() == (C // 64) * (nd + na) * 8 * 64
    # the builder's own float64 measurement of the truncation error between the nodes, relative to the amplitude
    assert fit[0] < 3e-8 * fit[1] and fit[2] < 3e-8 * fit[3], fit
    tab = table.numpy()
    rng = np.random.default_rng(8)
    xd, xa = rng.uniform(0, 48.0, 4000), rng.uniform(0, 12.0, 4000)
    xa[:2] = (0.0, 12.0)    # atan2 end points: 180 / sigma_a must lie inside the last interval
    for x, base, w, b, amp in ((xd, 0, wd, bd, fit[1]), (xa, nd, wa, ba, fit[3])):
        err = np.abs(_horner(tab, nd + na, x, h, base) - _exact(x, div, w, b)).max()
        assert err < 2.5e-7 * amp, (h, scale, base, err, amp)   # fp32 output rounding + seven fp32 Horner steps


def test_table_is_more_accurate_than_an_fp32_matmul():
    """The reason the table is the default: against float64 it is closer than the fp32 sinusoid + fp32 matmul it replaces."""
    from roitr_amd import ops
    div, wd, bd, wa, ba = _weights(9, 1.0 / 16)
    t = torch.from_numpy
    table, nd, na, fit = ops.geo_table_build(t(div), t(wd), t(bd), t(wa), t(ba), interval=2.0)
    x = np.random.default_rng(10).uniform(0, 20.0, 3000)
    ref = _exact(x, div, wd, bd)
    om = x.astype(np.float32)[:, None] * div[None]
    emb = np.empty((len(x), C), np.float32)
    emb[:, 0::2], emb[:, 1::2] = np.sin(om), np.cos(om)
    gemm = emb @ wd.T + bd
    e_tab = np.abs(_horner(table.numpy(), nd + na, x, 2.0, 0) - ref).max()
    e_gemm = np.abs(gemm - ref).max()
    assert e_tab < 0.5 * e_gemm, (e_tab, e_gemm)


def test_builder_rejects_bad_arguments():
    from roitr_amd import _lib as L
    import ctypes
    lib = L.lib()
    z = torch.zeros(C * C)
    fit = (ctypes.c_double * 6)()
    out = torch.zeros(int(lib.roitr_geo_table_floats(C, 4, 4)))
    hp = L.host_ptr   # the table builder is a host function
    args = lambda h, nd: (C, hp(z), hp(z), hp(z), hp(z), hp(z), L.c_float(h), nd, 4, hp(out), fit)
    assert lib.roitr_geo_table_build(*args(3.0, 4)) != 0     # the interval must be a power of two (exact index arithmetic)
    assert lib.roitr_geo_table_build(*args(2.0, 0)) != 0
    assert lib.roitr_geo_table_build(*args(2.0, 4)) == 0
