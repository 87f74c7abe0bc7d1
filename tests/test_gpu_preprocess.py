"""Preprocessing on the GPU against the oracle: workspace filter, voxel grid,
uniform sub-sampling -- GraspDetector::preprocessPointCloud, grasp_detector.cpp:285-335.
Integer/index work: every comparison is bit-exact."""
import numpy as np
import pytest

from conftest import scene_params
from agile_grasp2_amd import scene, weights

pytestmark = pytest.mark.gpu

WS = [0.0, 1.0, -0.5, 0.5, -0.2, 0.8]


def raw_cloud(seed, n, nan_every=97):
    rng = np.random.default_rng(seed)
    pts = (rng.uniform(-0.2, 1.2, size=(n, 3)) - [0, 0.5, 0.2]).astype(np.float32)
    pts[::nan_every, seed % 3] = np.nan
    pts[5::nan_every * 3] = np.inf
    return pts


def pair(**kw):
    from agile_grasp2_amd import capi
    from oracle import api
    return capi.Detector(**kw), api.Oracle(**kw)


@pytest.mark.parametrize("voxelize,voxel", [(False, 0.003), (True, 0.01), (True, 0.003)])
def test_filter_and_voxel_grid_match_oracle(voxelize, voxel):
    d, o = pair(workspace=WS)
    pts = raw_cloud(1, 60000)
    m_d = d.preprocess_cloud(pts, voxelize=voxelize, voxel_size=voxel)
    m_o = o.preprocess_cloud(pts, voxelize=voxelize, voxel_size=voxel)
    assert m_d == m_o and 0 < m_d < 60000
    xd, cd = d.get_cloud()
    xo, co = o.get_cloud()
    assert xd.tobytes() == xo.tobytes()
    assert np.array_equal(cd, co) and cd.min() == 1
    assert xd[:, 0].min() > WS[0] - voxel and xd[:, 0].max() < WS[1]
    if voxelize:  # independent numpy statement of cloud_camera.cpp:124-168
        keep = np.isfinite(pts).all(1) & (pts[:, 0] > WS[0]) & (pts[:, 0] < WS[1]) & \
            (pts[:, 1] > WS[2]) & (pts[:, 1] < WS[3]) & (pts[:, 2] > WS[4]) & (pts[:, 2] < WS[5])
        assert xd.tobytes() == scene.voxelize(pts[keep], voxel).tobytes()
    # the grid built over the processed cloud is the one ag2_set_cloud builds
    assert np.array_equal(d.get_grid_perm(), o.get_grid_perm())
    d.close()


def test_no_filter_keeps_everything_finite():
    d, o = pair(workspace=WS)
    pts = raw_cloud(2, 20000)
    assert d.preprocess_cloud(pts, filter_workspace=False, voxelize=False) == \
        o.preprocess_cloud(pts, filter_workspace=False, voxelize=False) == int(np.isfinite(pts).all(1).sum())
    assert d.get_cloud()[0].tobytes() == o.get_cloud()[0].tobytes()
    assert d.preprocess_cloud(pts, filter_workspace=False, voxelize=True, voxel_size=0.02) == \
        o.preprocess_cloud(pts, filter_workspace=False, voxelize=True, voxel_size=0.02)
    assert d.get_cloud()[0].tobytes() == o.get_cloud()[0].tobytes()
    d.close()


def test_points_on_voxel_boundaries():
    """Coordinates that are exact multiples of the cell from the minimum: the float division and
    floor must round the same way on both sides (and as numpy does)."""
    d, o = pair(workspace=[-10, 10, -10, 10, -10, 10])
    rng = np.random.default_rng(5)
    cell = np.float32(0.003)
    mn = np.array([0.1234, -0.777, 0.5], dtype=np.float32)
    ijk = rng.integers(0, 200, size=(50000, 3))
    pts = (ijk.astype(np.float32) * cell + mn).astype(np.float32)
    pts[0] = mn
    pts[1::2] = np.nextafter(pts[1::2], np.float32(np.inf))
    pts[2::4] = np.nextafter(pts[2::4], np.float32(-np.inf))
    pts[0] = mn
    assert d.preprocess_cloud(pts, voxel_size=float(cell)) == o.preprocess_cloud(pts, voxel_size=float(cell))
    xd = d.get_cloud()[0]
    assert xd.tobytes() == o.get_cloud()[0].tobytes() == scene.voxelize(pts, float(cell)).tobytes()
    d.close()


@pytest.mark.parametrize("flags", [0, 1])
def test_two_cameras_and_normals(flags):
    kw = dict(workspace=WS, n_cams=2, cam_origin=[[0, 0, 0], [0.3, 0, 0]])
    d, o = pair(**kw)
    pts = raw_cloud(3, 30000)
    rng = np.random.default_rng(4)
    cam = rng.integers(0, 2, size=(2, 30000)).astype(np.int32)
    cam[:, ::11] = 2  # anything but 1 counts as "not seen" (cloud_camera.cpp:151)
    assert d.preprocess_cloud(pts, cam_source=cam, voxel_size=0.01, flags=flags) == \
        o.preprocess_cloud(pts, cam_source=cam, voxel_size=0.01, flags=flags)
    (xd, cd), (xo, co) = d.get_cloud(), o.get_cloud()
    assert xd.tobytes() == xo.tobytes() and np.array_equal(cd, co)
    assert set(np.unique(cd)) <= {0, 1} and 0.1 < cd.mean() < 0.9
    # filter only, normals carried through (cloud_camera.cpp:109-117)
    nrm = rng.normal(size=(3, 30000))
    nrm /= np.linalg.norm(nrm, axis=0)
    m = d.preprocess_cloud(pts, cam_source=cam, normals=nrm, voxelize=False)
    assert m == o.preprocess_cloud(pts, cam_source=cam, normals=nrm, voxelize=False)
    (xd, cd), (xo, co) = d.get_cloud(), o.get_cloud()
    assert xd.tobytes() == xo.tobytes() and np.array_equal(cd, co)
    assert np.array_equal(d.get_normals(), o.get_normals())
    with pytest.raises(RuntimeError):
        d.preprocess_cloud(pts, cam_source=cam, normals=nrm, voxelize=True)
    d.close()


def test_empty_and_fully_filtered():
    d, o = pair(workspace=WS)
    assert d.preprocess_cloud(np.zeros((0, 3), np.float32)) == 0
    far = raw_cloud(1, 1000, nan_every=10 ** 9) + np.float32(50.0)
    assert d.preprocess_cloud(far) == o.preprocess_cloud(far) == 0
    assert d.subsample_uniformly(10).shape == (0,)
    one = np.array([[0.5, 0.0, 0.1]], np.float32)
    assert d.preprocess_cloud(one) == o.preprocess_cloud(one) == 1
    assert d.get_cloud()[0].tobytes() == o.get_cloud()[0].tobytes() == one.tobytes()
    d.close()


@pytest.mark.parametrize("n,k,seed", [(5000, 100, 0), (5000, 4999, 1), (5000, 5000, 2), (5000, 9000, 3),
                                      (200000, 5000, 4), (3, 1, 5), (70000, 1, 6), (70000, 69999, 7),
                                      # candidate lists of the one-workgroup selection (<= 8192 entries) and beyond
                                      (300000, 2000, 11), (200000, 6000, 10), (200000, 8000, 8), (200000, 20000, 12)])
def test_subsample_matches_oracle(n, k, seed):
    d, o = pair(workspace=[-10, 10, -10, 10, -10, 10])
    pts = np.random.default_rng(seed).uniform(-1, 1, size=(n, 3)).astype(np.float32)
    d.set_cloud(pts)
    o.set_cloud(pts)
    a, b = d.subsample_uniformly(k, seed=seed), o.subsample_uniformly(k, seed=seed)
    assert np.array_equal(a, b)
    assert len(a) == min(n, k) and np.all(np.diff(a) > 0) and (len(a) == 0 or (a[0] >= 0 and a[-1] < n))
    if k < n:  # a different seed gives a different draw
        assert not np.array_equal(a, d.subsample_uniformly(k, seed=seed + 100))
    d.close()


def test_subsample_is_uniform():
    from agile_grasp2_amd import capi
    d = capi.Detector(workspace=[-10, 10, -10, 10, -10, 10])
    n, k, reps = 4000, 400, 200
    d.set_cloud(np.random.default_rng(0).uniform(-1, 1, size=(n, 3)).astype(np.float32))
    hits = np.zeros(n)
    for s in range(reps):
        hits[d.subsample_uniformly(k, seed=s)] += 1
    # each index is drawn with probability k/n; binomial(reps, 0.1): mean 20, sd 4.2
    assert abs(hits.mean() - reps * k / n) < 1e-9
    assert hits.min() >= 2 and hits.max() <= 45
    quart = hits.reshape(4, -1).sum(axis=1)
    assert np.all(np.abs(quart - quart.mean()) < 5 * np.sqrt(quart.mean()))
    d.close()


def test_whole_front_end_then_detect():
    """raw cloud -> filter + voxel grid -> sub-sample -> normals -> detect, with the sample indices
    never leaving the device, equals the oracle run on the same raw cloud."""
    raw, ws = scene.make_scene(seed=6, n_target=60000, voxel=None, spacing=0.0015)
    kw = scene_params(ws, num_threads=4)
    d, o = pair(**kw)
    w = weights.make_lenet_weights(2)
    d.lenet_load(w)
    o.lenet_load(w)
    m = d.preprocess_cloud(raw)
    assert m == o.preprocess_cloud(raw) and 5000 < m < 40000
    ns = d.subsample_uniformly(150, seed=9, want_indices=False)
    idx = o.subsample_uniformly(150, seed=9)
    assert ns == 150 == len(idx)
    d.compute_normals()
    o.compute_normals()
    sel_d, all_d = d.detect(n_resident=ns, seed=11)
    sel_o, all_o = o.detect(sample_idx=idx, seed=11)
    assert len(all_d) == len(all_o) > 10
    for f in ("sample_slot", "orientation", "half_antipodal", "n_points", "width", "bottom", "top",
              "surface", "axis", "approach", "binormal"):
        assert np.array_equal(all_d[f], all_o[f]), f
    tol = 1e-4 * np.abs(all_o["score"]).max() + 2e-3
    assert np.abs(all_d["score"] - all_o["score"]).max() <= tol
    # explicit indices give the same bytes as the resident ones
    sel_e, all_e = d.detect(sample_idx=idx, seed=11)
    assert all_e.tobytes() == all_d.tobytes() and sel_e.tobytes() == sel_d.tobytes()
    assert d.times().preprocess_ms > 0
    d.close()


def test_device_resident_raw_cloud_and_full_size_properties():
    """BASELINE-config-2-sized front end: ~1.3 M raw points in HBM -> ~300 k voxels.  The oracle is
    too slow to be the checker here, numpy's unique is not: sorted, unique, inside the workspace,
    same bytes as the host entry point."""
    import ctypes as C
    from agile_grasp2_amd import capi
    raw, ws = scene.make_scene(seed=1, n_target=1300000, voxel=None, spacing=0.0015)
    d = capi.Detector(**scene_params(ws))
    hip = C.CDLL("libamdhip64.so.7")  # the runtime libag2hip.so already brought in (by soname)
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), C.c_size_t(raw.nbytes)) == 0
    assert hip.hipMemcpy(dptr, raw.ctypes.data_as(C.c_void_p), C.c_size_t(raw.nbytes), C.c_int(1)) == 0
    m = d.preprocess_cloud_device(dptr.value, raw.shape[0], 12)
    assert hip.hipFree(dptr) == 0
    xd = d.get_cloud()[0]
    ref = scene.voxelize(raw[np.isfinite(raw).all(1) & (raw[:, 0] > ws[0]) & (raw[:, 0] < ws[1]) &
                             (raw[:, 1] > ws[2]) & (raw[:, 1] < ws[3]) & (raw[:, 2] > ws[4]) &
                             (raw[:, 2] < ws[5])])
    assert m == ref.shape[0] and 200000 < m < 800000
    assert xd.tobytes() == ref.tobytes()
    assert d.preprocess_cloud(raw) == m and d.get_cloud()[0].tobytes() == ref.tobytes()
    idx = d.subsample_uniformly(5000, seed=1)
    assert len(idx) == 5000 and np.all(np.diff(idx) > 0)
    d.close()
