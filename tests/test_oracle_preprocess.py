"""Oracle preprocessing (CloudCamera::filterWorkspace / voxelizeCloud / subsampleUniformly,
cloud_camera.cpp:89-178) pinned against independent numpy statements of the same code.
The reference holds no fixture for this step either: parity unpinned, as for the rest of the path."""
import numpy as np

from agile_grasp2_amd import scene
from oracle import api

WS = [0.0, 1.0, -0.5, 0.5, -0.2, 0.8]


def raw_cloud(seed, n):
    rng = np.random.default_rng(seed)
    pts = (rng.uniform(-0.2, 1.2, size=(n, 3)) - [0, 0.5, 0.2]).astype(np.float32)
    pts[::97, seed % 3] = np.nan
    return pts


def inside(p, ws):
    return np.isfinite(p).all(1) & (p[:, 0] > ws[0]) & (p[:, 0] < ws[1]) & (p[:, 1] > ws[2]) & \
        (p[:, 1] < ws[3]) & (p[:, 2] > ws[4]) & (p[:, 2] < ws[5])


def test_filter_keeps_order_and_strictness():
    o = api.Oracle(workspace=WS)
    pts = raw_cloud(0, 5000)
    pts[10] = [WS[0], 0.0, 0.0]     # on the bound: strict comparison drops it (cloud_camera.cpp:94)
    pts[11] = [np.nextafter(np.float32(WS[0]), np.float32(1)), 0.0, 0.0]
    m = o.preprocess_cloud(pts, voxelize=False)
    keep = inside(pts, WS)
    assert not keep[10] and keep[11]
    xyz, cam = o.get_cloud()
    assert m == keep.sum() and xyz.tobytes() == pts[keep].tobytes() and cam.min() == cam.max() == 1


def test_voxel_grid_matches_numpy_and_set_semantics():
    o = api.Oracle(workspace=WS, n_cams=2, cam_origin=[[0, 0, 0], [0.3, 0, 0]])
    pts = raw_cloud(1, 20000)
    cam = np.random.default_rng(2).integers(0, 3, size=(2, 20000)).astype(np.int32)
    cell = 0.01
    keep = inside(pts, WS)
    p, c = pts[keep], cam[:, keep]
    mn = p.min(axis=0)
    v = np.floor((p - mn) / np.float32(cell)).astype(np.int64)
    uniq, first = np.unique(v, axis=0, return_index=True)   # lexicographic, first occurrence
    for flags in (0, 1):
        assert o.preprocess_cloud(pts, cam_source=cam, voxel_size=cell, flags=flags) == len(uniq)
        xyz, cs = o.get_cloud()
        assert xyz.tobytes() == scene.voxelize(p, cell).tobytes()
        src = first if flags else np.sort(first)  # literal reference: k-th voxel <- k-th first hit
        assert np.array_equal(cs, (c[:, src] == 1).astype(np.int32))


def test_subsample_definition():
    o = api.Oracle(workspace=[-9, 9, -9, 9, -9, 9])
    pts = np.random.default_rng(3).uniform(-1, 1, size=(3000, 3)).astype(np.float32)
    o.set_cloud(pts)
    a = o.subsample_uniformly(300, seed=5)
    assert len(a) == 300 and np.all(np.diff(a) > 0)
    assert np.array_equal(a, o.subsample_uniformly(300, seed=5))
    assert not np.array_equal(a, o.subsample_uniformly(300, seed=6))
    # nested: the 100 smallest keys are among the 300 smallest
    assert set(o.subsample_uniformly(100, seed=5)) <= set(a)
    assert np.array_equal(o.subsample_uniformly(5000, seed=5), np.arange(3000))  # grasp_detector.cpp:321-328
    hits = np.zeros(3000)
    for s in range(100):
        hits[o.subsample_uniformly(300, seed=s)] += 1
    assert abs(hits.mean() - 10.0) < 1e-9 and hits.max() <= 28 and hits.min() >= 0
    assert abs(hits[:1500].sum() - hits[1500:].sum()) < 5 * np.sqrt(15000)
