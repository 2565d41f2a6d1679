"""Closed-form, name-keyed parameter values (no checkpoint blobs travel with the repo).

The released RoITr checkpoints (reference README.md:44,112) are not available offline, and a
40 MB random blob is not a fixture.  Instead every parameter is a pure function of its
state_dict key and shape, so the golden-vector generator (which overwrites the *reference*
model's parameters, tests/golden/make_golden.py) and the MI355X engine (which regenerates the
same values on the GPU box from key names alone) agree bit-for-bit.

The generator is a counter-based integer hash (splitmix64 finaliser) evaluated with wrapping
uint64 numpy arithmetic -- independent of numpy's Generator streams, so it is stable across
numpy versions.  Magnitudes follow torch's default initialisers so activations stay O(1):
  2-D weight (out, in)      U(-1/sqrt(in), 1/sqrt(in))
  1-D '.weight' (LayerNorm) 1 + U(-0.1, 0.1)
  1-D '.bias'               U(-0.1, 0.1)
  0-D (OT alpha)            1 + U(-0.1, 0.1)
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def hashed_uniform(key, numel):
    """numel float64 values in [0, 1), a pure function of (key, position)."""
    seed = np.uint64((zlib.crc32(key.encode("utf-8")) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF)
    ctr = np.arange(numel, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64(ctr * np.uint64(0xD1342543DE82EF95) + seed)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def closed_form_param(key, shape):
    """float32 array for the parameter called `key` with `shape`."""
    shape = tuple(int(s) for s in shape)
    numel = int(np.prod(shape)) if len(shape) else 1
    u = hashed_uniform(key, numel) * 2.0 - 1.0  # [-1, 1)
    if len(shape) == 2:
        bound = 1.0 / np.sqrt(shape[1])
        v = u * bound
    elif len(shape) == 1 and key.endswith(".weight"):
        v = 1.0 + 0.1 * u
    elif len(shape) == 1:
        v = 0.1 * u
    elif len(shape) == 0:
        v = 1.0 + 0.1 * u
    else:
        v = u * 0.05
    return np.asarray(v, dtype=np.float64).reshape(shape).astype(np.float32)


def closed_form_state(layout):
    """layout: iterable of (key, shape) -> dict key -> float32 ndarray."""
    return {k: closed_form_param(k, s) for k, s in layout}
