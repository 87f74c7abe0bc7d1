"""Synthetic hand lists for the clustering tests: hands strung along a few handle-like lines (small
perpendicular and angular jitter, random axis sign) plus scattered outliers."""
import numpy as np


def make_hands(dtype, seed, n, n_lines=12, outlier_frac=0.3):
    rng = np.random.default_rng(seed)
    h = np.zeros(n, dtype=dtype)
    if n == 0:
        return h
    n_out = int(outlier_frac * n)
    line = rng.integers(0, n_lines, size=n)
    p0 = rng.uniform([0.4, -0.4, 0.0], [0.9, 0.4, 0.3], size=(n_lines, 3))
    a = rng.normal(size=(n_lines, 3))
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    t = rng.uniform(-0.06, 0.06, size=n)
    bottom = p0[line] + t[:, None] * a[line] + rng.normal(scale=0.002, size=(n, 3))
    axis = a[line] + rng.normal(scale=0.08, size=(n, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    axis *= rng.choice([-1.0, 1.0], size=(n, 1))
    out = rng.permutation(n)[:n_out]
    bottom[out] = rng.uniform([0.4, -0.4, 0.0], [0.9, 0.4, 0.3], size=(n_out, 3))
    approach = np.cross(axis, rng.normal(size=(n, 3)))
    approach /= np.linalg.norm(approach, axis=1, keepdims=True)
    h["axis"] = axis
    h["approach"] = approach
    h["binormal"] = np.cross(approach, axis)
    h["bottom"] = bottom
    h["top"] = bottom + 0.06 * approach
    h["surface"] = bottom + 0.02 * approach
    h["width"] = rng.uniform(0.03, 0.07, size=n)
    h["score"] = rng.uniform(-500, 1500, size=n)
    h["sample_slot"] = np.arange(n)
    h["orientation"] = rng.integers(0, 8, size=n)
    h["half_antipodal"] = 1
    h["full_antipodal"] = 1
    h["n_points"] = rng.integers(20, 400, size=n)
    return h


def numpy_clusters(h, min_inliers):
    """Vectorised, independent statement of handle_search.cpp:4-80 (remove_inliers = false)."""
    a, b, s = h["axis"], h["bottom"], h["score"]
    n = len(h)
    aligned = np.abs(a @ a.T) > np.cos(np.deg2rad(15.0))
    d = b[:, None, :] - b[None, :, :]
    mag = np.linalg.norm(d, axis=2) <= 0.05
    proj = d - np.einsum("ijk,ik->ij", d, a)[:, :, None] * a[:, None, :]
    pm = np.linalg.norm(proj, axis=2) <= 0.005
    inl = aligned & mag & pm & ~np.eye(n, dtype=bool)
    cnt = inl.sum(axis=1)
    keep = cnt >= min_inliers
    mean_b = (inl.astype(np.float64) @ b)[keep] / cnt[keep, None]
    out = h[keep].copy()
    delta = mean_b - out["bottom"]
    for f in ("surface", "bottom", "top"):
        out[f] = out[f] + delta
    out["score"] = (inl.astype(np.float64) @ s)[keep] / cnt[keep]
    return out, cnt
