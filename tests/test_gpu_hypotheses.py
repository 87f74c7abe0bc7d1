"""GPU parity, K2 + K3 + K4 + prune: local frames, hand sweep, point lists, prune flags and grasp
images, HIP (through the C-ABI) vs the oracle on the same seeded inputs.

Bars: labels / indices / counts / image bytes bit-exact.  Poses: the north-star tolerance is 1e-4;
because both sides execute the same IEEE op sequence the tests demand bit equality and would
report the max deviation if that ever failed.
"""
import numpy as np
import pytest

from conftest import scene_params
from agile_grasp2_amd import scene

pytestmark = pytest.mark.gpu

VEC = ("axis", "approach", "binormal", "surface", "bottom", "top")


def make_pair(xyz, ws, cam_source=None, normals=None, **kw):
    from agile_grasp2_amd import capi
    from oracle import api
    prm = scene_params(ws, **kw)
    o = api.Oracle(**dict(prm, num_threads=8))
    d = capi.Detector(**prm)
    for x in (o, d):
        x.set_cloud(xyz, cam_source=cam_source, normals=normals)
        if normals is None:
            x.compute_normals()
    return o, d


def assert_hyps_equal(got, want):
    assert len(got) == len(want)
    for f in ("sample_slot", "orientation", "half_antipodal", "full_antipodal", "n_points"):
        assert np.array_equal(got[f], want[f]), f
    for f in VEC + ("width",):
        if not np.array_equal(got[f], want[f]):
            dev = np.abs(got[f] - want[f]).max()
            raise AssertionError(f"{f}: not bit-identical, max |diff| = {dev:g} (north-star tol 1e-4)")


def check_lists_and_images(o, d, hyps, stride=1):
    n = len(hyps)
    for k in range(0, n, stride):
        p = int(hyps[k]["n_points"])
        gp, gn = d.hyp_points(k, p)
        wp, wn = o.hyp_points(k, p)
        assert np.array_equal(gp, wp) and np.array_equal(gn, wn, equal_nan=True), k
    gi = d.render_images(0, n)
    wi = o.render_images(0, n)
    assert np.array_equal(gi, wi)
    assert np.array_equal(d.prune(n), o.prune(n))
    return gi


def test_frames_bit_exact(small_scene):
    xyz, ws, idx = small_scene
    o, d = make_pair(xyz, ws)
    gf, gv = d.local_frames(sample_idx=idx, slot_base=5, seed=77)
    wf, wv = o.local_frames(sample_idx=idx, slot_base=5, seed=77)
    assert np.array_equal(gv, wv)
    assert np.array_equal(gf.view(np.uint64), wf.view(np.uint64))
    d.close()


@pytest.mark.parametrize("debug_flags", [0, 1])
def test_hypotheses_small_scene(small_scene, debug_flags):
    """debug_flags=1 visits every radius neighbour (exact K2 counter); 0 culls rows by the sphere
    and the crop slab first.  Both must reproduce the oracle exactly."""
    xyz, ws, idx = small_scene
    o, d = make_pair(xyz, ws, debug_flags=debug_flags)
    got = d.generate_hypotheses(sample_idx=idx, seed=5)
    want = o.generate_hypotheses(sample_idx=idx, seed=5)
    assert len(want) > 20
    assert_hyps_equal(got, want)
    imgs = check_lists_and_images(o, d, want)
    assert imgs.max() > 0
    gc, wc = d.counters(), o.counters()
    fields = ("n_frames", "n_hypotheses", "sum_kcrop", "sum_p") + (("sum_k2",) if debug_flags else ())
    for f in fields:
        assert getattr(gc, f) == getattr(wc, f), f
    d.close()


def test_hypotheses_xyz_samples_and_slot_base(small_scene):
    xyz, ws, idx = small_scene
    o, d = make_pair(xyz, ws)
    rng = np.random.default_rng(8)
    sx = (xyz[idx[:40]].astype(np.float64) + rng.normal(scale=0.004, size=(40, 3))).T
    sx[:, 3] = [10.0, 10.0, 10.0]   # far outside the grid: no neighbours
    sx[:, 4] = np.nan               # invalid sample
    got = d.generate_hypotheses(sample_xyz=sx, slot_base=1000, seed=6)
    want = o.generate_hypotheses(sample_xyz=sx, slot_base=1000, seed=6)
    assert len(want) > 3
    assert want["sample_slot"].min() >= 1000
    assert_hyps_equal(got, want)
    check_lists_and_images(o, d, want)
    d.close()


def test_two_cameras_and_given_normals():
    xyz, ws = scene.make_scene(seed=12, n_target=5000, kind="objects")
    n = xyz.shape[0]
    cam = np.zeros((2, n), dtype=np.int32)
    cam[0, : n // 3] = 1
    cam[1, n // 3:] = 1
    cams = [scene.CAMERA, scene.CAMERA + np.array([0.0, 0.6, 0.1])]
    o, d = make_pair(xyz, ws, cam_source=cam, n_cams=2, cam_origin=cams)
    idx = scene.draw_samples(2, n, 100)
    got = d.generate_hypotheses(sample_idx=idx, seed=1)
    want = o.generate_hypotheses(sample_idx=idx, seed=1)
    assert_hyps_equal(got, want)
    # externally supplied normals (CloudCamera with normals, cloud_camera.cpp:4-32)
    nrm = o.get_normals()
    o2, d2 = make_pair(xyz, ws, cam_source=cam, normals=nrm, n_cams=2, cam_origin=cams)
    got2 = d2.generate_hypotheses(sample_idx=idx, seed=1)
    want2 = o2.generate_hypotheses(sample_idx=idx, seed=1)
    assert_hyps_equal(got2, want2)
    assert_hyps_equal(got2, want)
    d.close()
    d2.close()


def test_sixteen_orientations_filter_half(small_scene):
    xyz, ws, idx = small_scene
    o, d = make_pair(xyz, ws, num_orientations=16, filter_half_grasps=1, init_bite=0.015)
    got = d.generate_hypotheses(sample_idx=idx[:80], seed=3)
    want = o.generate_hypotheses(sample_idx=idx[:80], seed=3)
    assert want["orientation"].max() > 8
    assert_hyps_equal(got, want)
    check_lists_and_images(o, d, want, stride=2)
    d.close()


@pytest.mark.parametrize("n_orient", [20, 32])
def test_more_than_sixteen_orientations(small_scene, n_orient):
    """17 .. 32 orientations run the sweep kernels' third instantiation (k_sweep<0, 32> / <1, 32>: per-orientation
    state in 32 register slots), on the LDS stage (the voxelised small scene) and on the long-list stage
    (a dense un-voxelised cloud)."""
    xyz, ws, idx = small_scene
    o, d = make_pair(xyz, ws, num_orientations=n_orient)
    got = d.generate_hypotheses(sample_idx=idx, seed=11)
    want = o.generate_hypotheses(sample_idx=idx, seed=11)
    assert len(want) > 10 and want["orientation"].max() >= 16
    assert_hyps_equal(got, want)
    assert got.tobytes() == want.tobytes()
    check_lists_and_images(o, d, want)
    d.close()
    dense, wsd = scene.make_scene(seed=4, n_target=120000, kind="objects", voxel=None)
    idd = scene.draw_samples(4, dense.shape[0], 40)
    o, d = make_pair(dense, wsd, num_orientations=n_orient)
    got = d.generate_hypotheses(sample_idx=idd, seed=12)
    want = o.generate_hypotheses(sample_idx=idd, seed=12)
    assert d.counters().n_overflow_samples > 0
    assert_hyps_equal(got, want)
    assert got.tobytes() == want.tobytes()
    d.close()


def test_config1_scene_500_samples():
    """BASELINE config 1 size: ~50k points, 500 samples, 8 orientations."""
    xyz, ws = scene.make_scene(seed=21, n_target=50000)
    idx = scene.draw_samples(21, xyz.shape[0], 500)
    o, d = make_pair(xyz, ws)
    got = d.generate_hypotheses(sample_idx=idx, seed=21)
    want = o.generate_hypotheses(sample_idx=idx, seed=21)
    assert len(want) > 100
    assert_hyps_equal(got, want)
    check_lists_and_images(o, d, want, stride=7)
    d.close()


@pytest.mark.parametrize("debug_flags", [0, 2])
def test_dense_unvoxelised_cloud_overflow_path(debug_flags):
    """Un-voxelised dense clutter (config 3 style): cropped neighbourhoods exceed the LDS stage,
    so the global-scratch instantiation of the sweep must give the same answers.  debug_flags=2
    starts that scratch at 1 024 points per workgroup: every list of the stage is then too long for
    it and the run is repeated with the scratch sized to the longest list."""
    xyz, ws = scene.make_scene(seed=4, n_target=120000, kind="objects", voxel=None)
    idx = scene.draw_samples(4, xyz.shape[0], 60)
    o, d = make_pair(xyz, ws, num_orientations=16, debug_flags=debug_flags)
    got = d.generate_hypotheses(sample_idx=idx, seed=9)
    want = o.generate_hypotheses(sample_idx=idx, seed=9)
    assert_hyps_equal(got, want)
    assert d.counters().n_overflow_samples > 0
    check_lists_and_images(o, d, want, stride=5)
    # a second run on the same context reuses the resized scratch
    got = d.generate_hypotheses(sample_idx=idx[::-1].copy(), seed=10)
    want = o.generate_hypotheses(sample_idx=idx[::-1].copy(), seed=10)
    assert_hyps_equal(got, want)
    d.close()


def test_cropped_list_longer_than_default_scratch():
    """hand_search.cpp:329-349 crops into a list of ANY length.  This scene (round 1's
    tools/fuzz_more.py third dense scene: 320 k un-voxelised tabletop points, 12 orientations) has
    samples whose cropped neighbourhood exceeds 65 536 points, the default size of the sweep's
    global scratch; round 1 returned AG2_ERR_CAPACITY for the whole call."""
    xyz, ws = scene.make_scene(seed=3, n_target=320000, kind="tabletop", voxel=None)
    idx = scene.draw_samples(3, xyz.shape[0], 120)
    o, d = make_pair(xyz, ws, num_orientations=12)
    got = d.generate_hypotheses(sample_idx=idx, seed=3)
    want = o.generate_hypotheses(sample_idx=idx, seed=3)
    assert len(want) > 50
    assert_hyps_equal(got, want)
    assert got.tobytes() == want.tobytes()
    check_lists_and_images(o, d, want, stride=3)
    d.close()


def test_very_dense_cloud_units_longer_than_a_wave():
    """Half-millimetre spacing: a neighbourhood holds well over 100 000 candidates, so the one-pass
    crop of the long-list stage works in units of more than 64 points (the unit table has 1 024
    entries) and a list spans all four of its segments many times over."""
    xyz, ws = scene.make_scene(seed=5, n_target=400000, kind="objects", voxel=None, spacing=0.0005)
    idx = scene.draw_samples(5, xyz.shape[0], 24)
    o, d = make_pair(xyz, ws, num_orientations=8)
    got = d.generate_hypotheses(sample_idx=idx, seed=5)
    want = o.generate_hypotheses(sample_idx=idx, seed=5)
    c = d.counters()
    assert c.sum_kcrop // max(1, c.n_frames) > 40000
    assert_hyps_equal(got, want)
    assert got.tobytes() == want.tobytes()
    check_lists_and_images(o, d, want, stride=2)
    d.close()


def test_images_from_points_edge_cases():
    from agile_grasp2_amd import capi
    from oracle import api
    o = api.Oracle()
    d = capi.Detector()
    rng = np.random.default_rng(3)
    lists = []
    # empty list, single NaN normal, aliasing x-cell >= 60, dense collisions, out-of-range cells
    lists.append((np.zeros((3, 0)), np.zeros((3, 0))))
    lists.append((np.array([[0.5], [0.0], [0.5]]), np.array([[np.nan], [0.0], [1.0]])))
    lists.append((np.array([[1.0 + 1e-9, 0.2], [0.0, 0.5], [0.5, 0.5]]), np.array([[1.0, 0.0], [0.0, 0.0], [0.0, 1.0]])))
    u = rng.uniform(0.15, 0.85, size=(3, 5000))
    u[:, :2000] = np.round(u[:, :2000] * 20) / 20.0
    lists.append((u, rng.normal(size=(3, 5000))))
    u2 = rng.uniform(-0.5, 1.5, size=(3, 300))
    lists.append((u2, rng.normal(size=(3, 300))))
    # the sparse renderer's limit (1024 points, many per cell: dozens of accumulation rounds; zero
    # normal sums) and the first size the dense renderer takes
    # ... the sorted renderer's two ranges (1025..4096 and ..16384, keys padded to a power of two)
    # and the first size beyond them
    for npts in (1023, 1024, 1025, 7, 64, 257, 2047, 2048, 2049, 3000, 4096, 4097, 9000, 16384, 16385):
        u3 = rng.uniform(0.2, 0.8, size=(3, npts))
        u3[:2, : npts // 2] = np.round(u3[:2, : npts // 2] * 6) / 6.0
        n3 = rng.normal(size=(3, npts))
        n3[:, 1::7] = -n3[:, 0:-1:7][:, : n3[:, 1::7].shape[1]]   # pairs that cancel where they share a cell
        if npts == 3000:  # dropped points (cells outside the image) and aliasing x-cells in between
            u3[0, 5::11] = 1.3
            u3[1, 3::13] = 2.5
        lists.append((u3, n3))
    got = d.render_images_from_points([a for a, _ in lists], [b for _, b in lists])
    for k, (a, b) in enumerate(lists):
        want = o.render_image_from_points(a, b) if a.shape[1] else np.zeros((60, 60, 3), np.uint8)
        assert np.array_equal(got[k], want), k
    d.close()


def test_export_candidates_is_the_fixed_slot_table(small_scene):
    """ag2_export_candidates_device: occupied slots carry their record, every other slot is zero --
    also after an earlier run left other records in the (never cleared) slot table."""
    import ctypes as C
    from agile_grasp2_amd import capi, sharding
    xyz, ws, idx = small_scene
    R = 8
    d = capi.Detector(**scene_params(ws))
    d.set_cloud(xyz)
    d.compute_normals()
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    nbytes = len(idx) * R * sharding.SLOT_BYTES
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), nbytes) == 0
    for samples, seed in ((idx, 3), (idx[::-1].copy(), 4), (idx[: len(idx) // 3], 5)):
        recs = d.generate_hypotheses(sample_idx=samples, slot_base=0, seed=seed)
        assert len(recs) > 3
        nb = len(samples) * R * sharding.SLOT_BYTES
        d.export_candidates_device(dptr.value, nb)
        assert hip.hipDeviceSynchronize() == 0
        got = np.zeros(len(samples) * R, dtype=capi.HYP_DTYPE)
        assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), dptr, nb, 2) == 0
        want = sharding.table_from_records(recs, 0, len(samples), R, len(samples))
        assert got.tobytes() == want.tobytes()
        # compact form: header + the occupied slots in slot order; a cap below the count cuts the list
        for cap in (len(recs) + 5, max(1, len(recs) // 2)):
            cb = sharding.compact_bytes(cap)
            assert cb <= nbytes
            d.export_candidates_compact_device(dptr.value, cb, cap)
            assert hip.hipDeviceSynchronize() == 0
            raw = np.zeros(cb, dtype=np.uint8)
            assert hip.hipMemcpy(raw.ctypes.data_as(C.c_void_p), dptr, cb, 2) == 0
            used = sharding.COMPACT_HEADER + min(cap, len(recs)) * sharding.SLOT_BYTES  # the rest is not written
            assert raw[:used].tobytes() == sharding.pack_compact(recs, cap)[:used].tobytes()
            out, cut = sharding.unpack_compact(raw, 1, cap, capi.HYP_DTYPE)
            assert cut == (cap < len(recs)) and out.tobytes() == recs[:cap].tobytes()
    hip.hipFree(dptr)
    d.close()
