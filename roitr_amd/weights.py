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


# ---- the "selective" variant --------------------------------------------------------------------------------------------
# With the plain initialiser-sized weights every descriptor is dominated by one common vector (constant input features,
# biases): all superpoint descriptors of a cloud agree to cos > 0.97, so the 4DMatch similarity threshold (0.75,
# model/RIGA_v2.py:27) admits EVERY node pair and no fine-matching score clears 0.05.  The selective variant keeps the same
# hash but (a) raises the gain of the geometry-dependent inputs -- the local PPF embeddings (x8), the two projections of the
# geometric structure embedding (x4) -- and of the point-descriptor head `fine_proj` (x4: the patch scores are quadratic in
# it), and (b) moves `coarse_proj.bias` by -W c and `fine_proj.bias` by -W p, where c / p are fixed per-width vectors
# (configs/selective_centre.npz: the mean global-transformer output resp. the mean input of fine_proj on one synthetic
# calibration pair of the bench size, c scaled so that a few percent of the node pairs fall under the threshold there;
# written by tests/golden/calibrate_selective.py) -- descriptors then vary around zero like a trained network's.  It is still a pure function of
# (key, shape) plus that committed constant, so the reference model, the oracle and the engine agree on it bit for bit.
SELECTIVE_GAIN = {".embedding.proj.weight": 8.0, ".embedding.proj_d.weight": 4.0, ".embedding.proj_a.weight": 4.0,
                  "fine_proj.weight": 4.0, "fine_proj.bias": 4.0}
_CENTRE = None


def selective_centre(name):
    """A committed centring vector: 'c256' / 'c512' (input of coarse_proj, 3DMatch / 4DMatch widths), 'p64' / 'p128' (input of
    fine_proj)."""
    global _CENTRE
    if _CENTRE is None:
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", "selective_centre.npz")
        with np.load(path) as z:
            _CENTRE = {k: z[k].astype(np.float64) for k in z.files}
    return np.asarray(_CENTRE[name], np.float64)


def closed_form_param(key, shape, variant="plain"):
    """float32 array for the parameter called `key` with `shape`; variant: 'plain' | 'selective' (see above)."""
    if variant == "selective":
        base = closed_form_param(key, shape).astype(np.float64)
        if key == "coarse_proj.bias":
            n = int(shape[0])
            w = closed_form_param("coarse_proj.weight", (n, n)).astype(np.float64)
            base = base - w @ selective_centre(f"c{n}")
        if key == "fine_proj.bias":
            n = int(shape[0])
            w = closed_form_param("fine_proj.weight", (n, n // 4)).astype(np.float64)
            base = base - w @ selective_centre(f"p{n // 4}")
        for suffix, gain in SELECTIVE_GAIN.items():
            if key.endswith(suffix):
                base = base * gain
        return base.astype(np.float32)
    if variant != "plain":
        raise ValueError(f"unknown weight variant {variant!r}")
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


def closed_form_state(layout, variant="plain"):
    """layout: iterable of (key, shape) -> dict key -> float32 ndarray."""
    return {k: closed_form_param(k, s, variant) for k, s in layout}
