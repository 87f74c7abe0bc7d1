"""GPU parity, K0 + K1: search grid order and PCA normals, HIP (through the C-ABI) vs the oracle.

Bars: grid permutation is index work -> bit-exact; normals are float results of an identical IEEE
op sequence -> bit-exact (compared as raw bits, NaN mask included).
"""
import numpy as np
import pytest

from conftest import scene_params
from agile_grasp2_amd import scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det_small(small_scene):
    from agile_grasp2_amd import capi
    xyz, ws, idx = small_scene
    d = capi.Detector(**scene_params(ws))
    d.set_cloud(xyz)
    d.compute_normals()
    yield d
    d.close()


def test_grid_perm_matches_oracle(small_scene, oracle_small, det_small):
    got = det_small.get_grid_perm()
    want = oracle_small.get_grid_perm()
    assert np.array_equal(got, want)


def test_normals_bit_exact(small_scene, oracle_small, det_small):
    got = det_small.get_normals().astype(np.float32)
    want = oracle_small.get_normals().astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    c = det_small.counters()
    assert c.sum_k1 == oracle_small.counters().sum_k1


def test_normals_with_invalid_and_sparse_points():
    from agile_grasp2_amd import capi
    from oracle import api
    rng = np.random.default_rng(5)
    xyz, ws = scene.make_scene(seed=9, n_target=3000, kind="objects")
    xyz = xyz.copy()
    xyz[rng.choice(xyz.shape[0], 40, replace=False)] = np.nan          # invalid points
    xyz = np.concatenate([xyz, np.array([[5.0, 5.0, 5.0], [-3.0, 2.0, 1.0]], dtype=np.float32)])  # isolated
    # pcl::PointXYZRGBA-like 32-byte stride
    wide = np.zeros((xyz.shape[0], 8), dtype=np.float32)
    wide[:, :3] = xyz
    o = api.Oracle(**scene_params(ws))
    o.set_cloud(wide[:, :3])
    o.compute_normals()
    d = capi.Detector(**scene_params(ws))
    d.set_cloud(wide[:, :3])
    d.compute_normals()
    nv = np.isfinite(xyz).all(axis=1).sum()
    assert np.array_equal(d.get_grid_perm(), o.get_grid_perm()[:nv])
    g, w = d.get_normals().astype(np.float32), o.get_normals().astype(np.float32)
    assert np.array_equal(np.isnan(g), np.isnan(w))
    assert np.array_equal(g[~np.isnan(g)].view(np.uint32), w[~np.isnan(w)].view(np.uint32))
    assert np.isnan(g[:, -1]).all() and np.isnan(g[:, -2]).all()
    d.close()


def test_empty_and_tiny_clouds():
    from agile_grasp2_amd import capi
    d = capi.Detector()
    d.set_cloud(np.zeros((0, 3), dtype=np.float32))
    d.compute_normals()
    assert d.get_normals().shape == (3, 0)
    d.set_cloud(np.array([[0.1, 0.2, 0.3]], dtype=np.float32))
    d.compute_normals()
    assert np.isnan(d.get_normals()).all()
    d.close()


def test_larger_cloud_bit_exact():
    """50k-point tabletop scene (BASELINE config 1 size): perm and normals bit-exact."""
    from agile_grasp2_amd import capi
    from oracle import api
    xyz, ws = scene.make_scene(seed=21, n_target=50000)
    o = api.Oracle(**scene_params(ws, num_threads=8))
    o.set_cloud(xyz)
    o.compute_normals()
    d = capi.Detector(**scene_params(ws))
    d.set_cloud(xyz)
    d.compute_normals()
    assert np.array_equal(d.get_grid_perm(), o.get_grid_perm())
    g, w = d.get_normals().astype(np.float32), o.get_normals().astype(np.float32)
    assert np.array_equal(g.view(np.uint32), w.view(np.uint32))
    d.close()


def test_cells_of_every_density():
    """Cell sort paths: many cells per LDS batch, several batches per workgroup (hundreds of points
    per cell), and one cell larger than the whole LDS stage (> 3072 points: sorted in place); device
    clouds (fused pack + extent pass) give the same grid as uploaded ones."""
    import ctypes as C
    from agile_grasp2_amd import capi
    from oracle import api
    rng = np.random.default_rng(17)
    base, ws = scene.make_scene(seed=2, n_target=4000)
    lo = base.min(axis=0)
    clump = (lo + np.float32(0.203) + rng.uniform(0, 0.004, size=(3500, 3))).astype(np.float32)   # one cell
    dense = (lo + np.float32(0.05) + rng.uniform(0, 0.06, size=(60000, 3))).astype(np.float32)     # ~280 per cell
    xyz = np.concatenate([base, clump, dense]).astype(np.float32)
    xyz = xyz[rng.permutation(xyz.shape[0])]
    o = api.Oracle(**scene_params(ws, num_threads=8))
    o.set_cloud(xyz)
    want = o.get_grid_perm()
    d = capi.Detector(**scene_params(ws))
    d.set_cloud(xyz)
    assert np.array_equal(d.get_grid_perm(), want)
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), xyz.nbytes) == 0
    assert hip.hipMemcpy(dptr, xyz.ctypes.data_as(C.c_void_p), xyz.nbytes, 1) == 0
    d.set_cloud_device(dptr.value, xyz.shape[0], 12)
    assert np.array_equal(d.get_grid_perm(), want)
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipFree(dptr)
    d.close()
