"""Synthetic 3DMatch-shaped pairs (there are no datasets in this image).

Input contract of the reference's collate (dataset/common.py:50-126, dataset/tdmatch.py:50-135):
float32 `src_points (N,3)`, `tgt_points (M,3)`, unit normals flipped toward the view point
(dataset/common.py:312-320 `normal_redirect`, view point = origin), `feats = ones (.,1)`
(dataset/tdmatch.py:128-129), `rot (3,3)`, `trans (3,1)` with  tgt ~= src @ rot.T + trans.T,
`raw_src_pcd = src_points` for 3DMatch.

Seed rule: numpy default_rng(1000 * config + pair_index).  Points ~ U[0,2)^3 m
(3DMatch fragments are 2-3 m across).  The two clouds are overlapping crops of one scene so the
matching stages have real correspondences to find.
"""
import numpy as np


def random_rotation(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def _normals(rng, pts):
    n = rng.standard_normal(pts.shape)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    flip = np.sum((0.0 - pts) * n, axis=1) < 0.0
    n[flip] *= -1.0
    return n


def field_normals(scene_pts):
    """A smooth unit vector field of the SCENE position: the same physical point carries the same normal in both clouds of a
    pair (up to the pair's rotation), so the PPFs of re-observed neighbourhoods agree -- what real surfaces give the reference."""
    p = scene_pts
    g = np.stack([np.cos(3.1 * p[:, 0] + 0.3) + 0.7 * np.sin(2.3 * p[:, 1]),
                  np.sin(2.7 * p[:, 1] + 1.1) + 0.6 * np.cos(3.3 * p[:, 2]),
                  np.cos(2.9 * p[:, 2] - 0.4) + 0.8 * np.sin(2.1 * p[:, 0] + p[:, 1])], 1) + 0.35
    return g / np.linalg.norm(g, axis=1, keepdims=True)


def euler_zyx(a):
    """scipy Rotation.from_euler('zyx', a).as_matrix() (dataset/tdmatch.py:102): extrinsic rotations about z, then y, then x."""
    cz, sz, cy, sy, cx, sx = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]])
    ry = np.array([[cy, 0, sy], [0, 1.0, 0], [-sy, 0, cy]])
    rx = np.array([[1.0, 0, 0], [0, cx, -sx], [0, sx, cx]])
    return rx @ ry @ rz


def surface_points(rng, n, x_lo, x_hi, noise=0.003):
    """n points on the surfaces of a room-like scene between x_lo and x_hi (round 4: scan-like geometry -- 3DMatch fragments are
    2-D surfaces, dataset/tdmatch.py:50-135 loads depth-fused scans): floor, ceiling and the two long walls of a 2 x 2 corridor
    along x, a wall across at every end, and box-shaped furniture standing on the floor; area-proportional sampling, Gaussian
    sensor noise.  Deterministic in rng."""
    L = x_hi - x_lo
    # axis-aligned rectangles: (origin, edge u, edge v)
    rects = [((x_lo, 0, 0), (L, 0, 0), (0, 2, 0)), ((x_lo, 0, 2), (L, 0, 0), (0, 2, 0)),        # floor, ceiling
             ((x_lo, 0, 0), (L, 0, 0), (0, 0, 2)), ((x_lo, 2, 0), (L, 0, 0), (0, 0, 2)),        # long walls
             ((x_lo, 0, 0), (0, 2, 0), (0, 0, 2)), ((x_hi, 0, 0), (0, 2, 0), (0, 0, 2))]        # end walls
    # furniture: boxes on a fixed lattice of the corridor (the same physical boxes whatever window [x_lo, x_hi] looks at them)
    for cx in np.arange(np.floor(x_lo / 0.9) * 0.9, x_hi + 0.9, 0.9):
        for cy, w, d, hgt in ((0.45, 0.5, 0.4, 0.8), (1.5, 0.35, 0.6, 0.45)):
            x0, y0 = cx + 0.15, cy - d / 2
            if x0 + w <= x_lo or x0 >= x_hi:
                continue
            xa, xb = max(x0, x_lo), min(x0 + w, x_hi)
            rects += [((xa, y0, hgt), (xb - xa, 0, 0), (0, d, 0)), ((xa, y0, 0), (xb - xa, 0, 0), (0, 0, hgt)),
                      ((xa, y0 + d, 0), (xb - xa, 0, 0), (0, 0, hgt))]
            if x0 >= x_lo:
                rects.append(((x0, y0, 0), (0, d, 0), (0, 0, hgt)))
            if x0 + w <= x_hi:
                rects.append(((x0 + w, y0, 0), (0, d, 0), (0, 0, hgt)))
    o = np.array([r[0] for r in rects], float); u = np.array([r[1] for r in rects], float); v = np.array([r[2] for r in rects], float)
    area = np.linalg.norm(np.cross(u, v), axis=1)
    pick = rng.choice(len(rects), size=n, p=area / area.sum())
    a, b = rng.random((n, 1)), rng.random((n, 1))
    return o[pick] + a * u[pick] + b * v[pick] + rng.normal(0.0, noise, (n, 3))


def make_pair(n_src, n_tgt=None, config=2, pair_index=0, overlap=0.6, jitter=0.002, rotated=None, normals="random", cloud="uniform"):
    """Returns a dict of float32 numpy arrays following the reference input contract.
    normals: 'random' (independent unit vectors per cloud, flipped toward the origin like dataset/common.py:312-320) or 'field'
    (field_normals of the scene position, carried through the pair's rigid transform: corresponding points share their normal).
    rotated (default: config == 3, BASELINE config 3 "3DLoMatch rotated"): the test-time rotation of dataset/tdmatch.py:99-112 --
    a seeded full-range euler rotation applied to the source or the target cloud, folded into rot / trans.
    cloud: 'uniform' (points ~ U[0,2)^3) or 'surface' (surface_points: piecewise-planar room-like scans with sensor noise)."""
    n_tgt = n_src if n_tgt is None else n_tgt
    rng = np.random.default_rng(1000 * config + pair_index)
    # one scene, two crops along x that share `overlap` of their extent
    shift = 2.0 * (1.0 - overlap)
    if cloud == "surface":
        src = surface_points(rng, n_src, 0.0, 2.0)
        tgt_scene = surface_points(rng, n_tgt, shift, 2.0 + shift)
    elif cloud == "uniform":
        src = rng.random((n_src, 3)) * 2.0
        tgt_scene = rng.random((n_tgt, 3)) * 2.0
        tgt_scene[:, 0] += shift
    else:
        raise ValueError(f"cloud must be 'uniform' or 'surface', got {cloud!r}")
    # inside the shared slab, tgt re-observes jittered src points (real correspondences)
    shared_src = np.nonzero(src[:, 0] >= shift)[0]
    shared_tgt = np.nonzero(tgt_scene[:, 0] < 2.0)[0]
    k = min(len(shared_src), len(shared_tgt))
    if k > 0:
        pick = rng.permutation(shared_src)[:k]
        tgt_scene[shared_tgt[:k]] = src[pick] + rng.normal(0.0, jitter, (k, 3))
    rot = random_rotation(rng)
    trans = rng.uniform(-1.0, 1.0, (3, 1))
    tgt = tgt_scene @ rot.T + trans.T
    src_n = tgt_n = None
    if normals == "field":
        src_n, tgt_n = field_normals(src), field_normals(tgt_scene) @ rot.T
    elif normals != "random":
        raise ValueError(f"normals must be 'random' or 'field', got {normals!r}")
    if rotated is None:
        rotated = config == 3
    if rotated:
        rot_ab = euler_zyx(rng.random(3) * np.pi * 2.0)
        if rng.random() > 0.5:
            src = src @ rot_ab.T
            rot = rot @ rot_ab.T
            src_n = src_n @ rot_ab.T if src_n is not None else None
        else:
            tgt = tgt @ rot_ab.T
            rot = rot_ab @ rot
            trans = rot_ab @ trans
            tgt_n = tgt_n @ rot_ab.T if tgt_n is not None else None
    if normals == "random":
        src_n = _normals(rng, src)
        tgt_n = _normals(rng, tgt)
    f32 = np.float32
    return {
        "src_points": np.ascontiguousarray(src, f32),
        "tgt_points": np.ascontiguousarray(tgt, f32),
        "src_normals": np.ascontiguousarray(src_n, f32),
        "tgt_normals": np.ascontiguousarray(tgt_n, f32),
        "src_feats": np.ones((n_src, 1), f32),
        "tgt_feats": np.ones((n_tgt, 1), f32),
        "rot": np.ascontiguousarray(rot, f32),
        "trans": np.ascontiguousarray(trans, f32),
        "raw_src_pcd": np.ascontiguousarray(src, f32),
    }
