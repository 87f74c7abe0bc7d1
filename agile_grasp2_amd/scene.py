"""Synthetic tabletop scenes for the agile_grasp2 hot path.

The reference ships no point cloud (launch/file_detect_grasps.launch:7 points at the author's home
directory), so every input here is synthetic: a table plane plus random boxes, cylinders and
spheres, surface-sampled on the camera-visible side only, with Gaussian noise, then snapped to the
reference's voxel grid exactly as CloudCamera::voxelizeCloud does
(src/agile_grasp2/cloud_camera.cpp:124-168: voxel value = floor((p - min) / cell) * cell + min in
float, output sorted lexicographically by (ix, iy, iz)).

Pure numpy; used by tests/ and bench.py to make inputs.  Not on the product's compute path.
"""
from __future__ import annotations

import numpy as np

VOXEL = 0.003  # grasp_detector.cpp:15
CAMERA = np.array([0.215, -0.015, 0.23])  # launch/robot_detect_grasps.launch:18-21 (translation)
TABLE_Z = -0.2


def voxelize(points: np.ndarray, cell: float = VOXEL) -> np.ndarray:
    """CloudCamera::voxelizeCloud (cloud_camera.cpp:124-168) in float32 arithmetic."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    if pts.shape[0] == 0:
        return pts
    mn = pts.min(axis=0)
    cellf = np.float32(cell)
    v = np.floor((pts - mn) / cellf).astype(np.int64)
    v = np.unique(v, axis=0)  # lexicographic (ix, iy, iz), like the std::set comparator
    out = v.astype(np.float32) * cellf + mn
    return np.ascontiguousarray(out, dtype=np.float32)


def _visible(p: np.ndarray, n: np.ndarray, cam: np.ndarray) -> np.ndarray:
    return np.einsum("ij,ij->i", n, cam[None, :] - p) > 0.0


def _box(rng, centre, size, yaw, density):
    sx, sy, sz = size
    faces = []
    areas = [sy * sz, sy * sz, sx * sz, sx * sz, sx * sy]
    axes = [(0, +1), (0, -1), (1, +1), (1, -1), (2, +1)]  # no bottom face
    for (ax, sgn), area in zip(axes, areas):
        m = max(int(area * density), 8)
        u = rng.uniform(-0.5, 0.5, size=(m, 3)) * np.array(size)
        u[:, ax] = sgn * 0.5 * size[ax]
        nn = np.zeros((m, 3))
        nn[:, ax] = sgn
        faces.append((u, nn))
    p = np.concatenate([f[0] for f in faces])
    n = np.concatenate([f[1] for f in faces])
    c, s = np.cos(yaw), np.sin(yaw)
    rot = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    return p @ rot.T + centre, n @ rot.T


def _cylinder(rng, centre, radius, height, lying, yaw, density):
    m = max(int(2 * np.pi * radius * height * density), 16)
    a = rng.uniform(0, 2 * np.pi, m)
    h = rng.uniform(-0.5, 0.5, m) * height
    p = np.stack([radius * np.cos(a), radius * np.sin(a), h], axis=1)
    n = np.stack([np.cos(a), np.sin(a), np.zeros(m)], axis=1)
    mc = max(int(np.pi * radius * radius * density), 8)
    for sgn in (+1.0, -1.0):
        rr = radius * np.sqrt(rng.uniform(0, 1, mc))
        aa = rng.uniform(0, 2 * np.pi, mc)
        pc = np.stack([rr * np.cos(aa), rr * np.sin(aa), np.full(mc, sgn * 0.5 * height)], axis=1)
        nc = np.zeros((mc, 3))
        nc[:, 2] = sgn
        p = np.concatenate([p, pc])
        n = np.concatenate([n, nc])
    if lying:
        rx = np.array([[1.0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])
        p, n = p @ rx.T, n @ rx.T
    c, s = np.cos(yaw), np.sin(yaw)
    rot = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    return p @ rot.T + centre, n @ rot.T


def _sphere(rng, centre, radius, density):
    m = max(int(4 * np.pi * radius * radius * density), 16)
    v = rng.normal(size=(m, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v * radius + centre, v


def make_scene(seed: int, n_target: int, kind: str = "tabletop", voxel: float | None = VOXEL,
               noise: float = 0.001, spacing: float = 0.001):
    """Return (xyz float32 [N,3], workspace[6]).  N is within about 1 % of n_target.

    kind: "tabletop" (table + clutter), "objects" (no table), "plane" (table only).
    voxel=None skips voxelisation (BASELINE config 3: dense un-voxelised clutter); the raw
    samples then have a mean spacing of `spacing` metres.
    """
    rng = np.random.default_rng(seed)
    cell = spacing if voxel is None else voxel
    per_pt = cell * cell  # surface area per voxelised point
    density = (4.0 if voxel is not None else 1.05) / per_pt  # raw samples per m^2
    table_frac = {"tabletop": 0.55, "objects": 0.0, "plane": 1.0}[kind]
    area_table = table_frac * n_target * per_pt
    asp = 1.5
    wx = max(np.sqrt(max(area_table, 1e-4) / asp), 0.25)
    wy = wx * asp
    x0, y0 = 0.48, -0.5 * wy
    pts = []
    if table_frac > 0:
        m = int(area_table * density)
        t = np.stack([rng.uniform(x0, x0 + wx, m), rng.uniform(y0, y0 + wy, m),
                      np.full(m, TABLE_Z)], axis=1)
        pts.append(t)

    def finish(chunks):
        p = np.concatenate(chunks)
        p = p + rng.normal(scale=noise, size=p.shape)
        p = p.astype(np.float32)
        return voxelize(p, voxel) if voxel is not None else p

    if kind != "plane":
        # add primitives until the (voxelised) count reaches the target
        est = int(area_table / per_pt)
        guard = 0
        while est < n_target and guard < 100000:
            guard += 1
            cx = rng.uniform(x0 + 0.05, x0 + wx - 0.05)
            cy = rng.uniform(y0 + 0.05, y0 + wy - 0.05)
            yaw = rng.uniform(0, np.pi)
            k = rng.integers(0, 4)
            if k == 0:
                size = rng.uniform(0.03, 0.08, 3)
                p, n = _box(rng, np.array([cx, cy, TABLE_Z + 0.5 * size[2]]), size, yaw, density)
            elif k == 1:
                r, h = rng.uniform(0.015, 0.04), rng.uniform(0.05, 0.2)
                p, n = _cylinder(rng, np.array([cx, cy, TABLE_Z + 0.5 * h]), r, h, False, yaw, density)
            elif k == 2:
                r, h = rng.uniform(0.015, 0.04), rng.uniform(0.05, 0.2)
                p, n = _cylinder(rng, np.array([cx, cy, TABLE_Z + r]), r, h, True, yaw, density)
            else:
                r = rng.uniform(0.02, 0.04)
                p, n = _sphere(rng, np.array([cx, cy, TABLE_Z + r]), r, density)
            vis = _visible(p, n, CAMERA)
            p = p[vis]
            pts.append(p)
            est += int(p.shape[0] / density / per_pt * 0.9)
    cloud = finish(pts)
    # trim or accept: keep a deterministic prefix-free subset if we overshoot by > 1 %
    if cloud.shape[0] > int(1.01 * n_target):
        keep = np.sort(rng.choice(cloud.shape[0], size=n_target, replace=False))
        cloud = cloud[keep]
    ws = np.array([x0 - 0.01, x0 + wx + 0.01, y0 - 0.01, y0 + wy + 0.01, TABLE_Z - 0.05, 1.0])
    return np.ascontiguousarray(cloud, dtype=np.float32), ws


def raw_from_voxels(vox: np.ndarray, seed: int, per_voxel: float = 2.55, cell: float = VOXEL) -> np.ndarray:
    """A raw "sensor" cloud whose voxelisation (CloudCamera::voxelizeCloud, cloud_camera.cpp:124-168) is the
    voxel cloud `vox` again: two or three points per voxel (per_voxel on average), each well inside its voxel,
    in random order; on every axis one point of a voxel with index 0 sits exactly on the lattice origin so that
    the minimum -- the origin the voxel indices are counted from -- is the same.  Lets a run WITH the front end
    be compared with the run on `vox` itself."""
    rng = np.random.default_rng(seed)
    vox = np.ascontiguousarray(vox, dtype=np.float32)
    mn = vox.min(axis=0)
    cellf = np.float32(cell)
    idx = np.rint((vox - mn) / cellf).astype(np.int64)
    reps = np.where(rng.uniform(size=len(vox)) < per_voxel - 2.0, 3, 2)
    owner = np.repeat(np.arange(len(vox)), reps)
    u = rng.uniform(0.25, 0.75, size=(len(owner), 3))
    pts = mn.astype(np.float64) + (idx[owner] + u) * float(cellf)
    for a in range(3):
        j = np.flatnonzero(idx[owner][:, a] == 0)[0]
        pts[j, a] = float(mn[a])
    raw = pts.astype(np.float32)
    return np.ascontiguousarray(raw[rng.permutation(len(raw))])


def draw_samples(seed: int, n_points: int, num_samples: int) -> np.ndarray:
    """Seeded stand-in for pcl::RandomSample (cloud_camera.cpp:171-178): num_samples indices
    without replacement, ascending."""
    rng = np.random.default_rng(seed ^ 0x5A17)
    k = min(num_samples, n_points)
    return np.sort(rng.choice(n_points, size=k, replace=False)).astype(np.int32)


def make_stream(seed: int, n_target: int, n_frames: int, motion: float = 0.004, voxel: float | None = VOXEL,
                noise: float = 0.001, spacing: float = 0.0015):
    """BASELINE.json configuration 5: n_frames clouds of ONE tabletop scene whose objects drift by
    about `motion` metres per frame (each object its own direction in the table plane), seen by the
    same camera with fresh sensor noise every frame, voxelised like make_scene.  Returns
    (list of xyz float32 [N_k, 3], workspace[6]); N_k is within one per cent of n_target.
    voxel=None: the RAW frames of the sensor (mean sample spacing `spacing`, n_target raw points) -- what
    ag2_detect_frame_raw takes; 765 000 raw points at 1.5 mm give about 300 000 voxels of 3 mm."""
    rng = np.random.default_rng(seed)
    cell = spacing if voxel is None else voxel
    per_pt = cell * cell
    density = (4.0 if voxel is not None else 1.05) / per_pt
    area_table = 0.55 * n_target * per_pt
    asp = 1.5
    wx = max(np.sqrt(max(area_table, 1e-4) / asp), 0.25)
    wy = wx * asp
    x0, y0 = 0.48, -0.5 * wy
    m = int(area_table * density)
    table = np.stack([rng.uniform(x0, x0 + wx, m), rng.uniform(y0, y0 + wy, m), np.full(m, TABLE_Z)], axis=1)
    objs, est, guard = [], int(area_table / per_pt), 0
    while est < n_target and guard < 100000:
        guard += 1
        cx = rng.uniform(x0 + 0.08, x0 + wx - 0.08)
        cy = rng.uniform(y0 + 0.08, y0 + wy - 0.08)
        yaw = rng.uniform(0, np.pi)
        k = int(rng.integers(0, 4))
        dims = rng.uniform(0.0, 1.0, 3)
        ang = rng.uniform(0, 2 * np.pi)
        objs.append((k, cx, cy, yaw, dims, np.array([np.cos(ang), np.sin(ang), 0.0]), int(rng.integers(1 << 30))))
        # visible area of the primitive, roughly (as in make_scene: counted after the visibility cut)
        p, _ = _make_object(objs[-1], 0.0, density)
        est += int(p.shape[0] / density / per_pt * 0.9)
    clouds = []
    for f in range(n_frames):
        chunks = [table]
        for o in objs:
            p, _ = _make_object(o, f * motion, density)
            chunks.append(p)
        p = np.concatenate(chunks)
        frng = np.random.default_rng([seed, f, 7])
        p = p + frng.normal(scale=noise, size=p.shape)
        cloud = voxelize(p.astype(np.float32), voxel) if voxel is not None else p.astype(np.float32)
        # as make_scene: a random subset brings the count to the target -- here the target +- 0.7 %,
        # a different count every frame
        want = n_target + (f * 7919) % (n_target // 70 + 1) - n_target // 140
        if cloud.shape[0] > want:
            cloud = cloud[np.sort(frng.choice(cloud.shape[0], size=want, replace=False))]
        clouds.append(np.ascontiguousarray(cloud, dtype=np.float32))
    ws = np.array([x0 - 0.01, x0 + wx + 0.01, y0 - 0.01, y0 + wy + 0.01, TABLE_Z - 0.05, 1.0])
    return clouds, ws


def _make_object(o, shift: float, density: float):
    """One primitive of make_stream, moved by `shift` along its drift direction, camera-visible side."""
    k, cx, cy, yaw, dims, direction, oseed = o
    rng = np.random.default_rng(oseed)  # the same surface samples in every frame: a rigid object
    c = np.array([cx, cy, 0.0]) + shift * direction
    if k == 0:
        size = 0.03 + 0.05 * dims
        p, n = _box(rng, np.array([c[0], c[1], TABLE_Z + 0.5 * size[2]]), size, yaw, density)
    elif k in (1, 2):
        r, h = 0.015 + 0.025 * dims[0], 0.05 + 0.15 * dims[1]
        zc = TABLE_Z + (0.5 * h if k == 1 else r)
        p, n = _cylinder(rng, np.array([c[0], c[1], zc]), r, h, k == 2, yaw, density)
    else:
        r = 0.02 + 0.02 * dims[0]
        p, n = _sphere(rng, np.array([c[0], c[1], TABLE_Z + r]), r, density)
    vis = _visible(p, n, CAMERA)
    return p[vis], n[vis]
