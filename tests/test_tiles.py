"""Spatial tiles ("tiling second"): a rank that holds only its samples'
x-interval plus a halo, binned against the whole cloud's minimum (ag2_set_grid_origin), must
produce exactly the hypotheses -- and prune decisions -- of the unsplit run, for any number of
ranks.  CPU: the oracle stands in for the per-rank compute; GPU: the HIP path through the C-ABI,
tiles run one after the other in one process."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from agile_grasp2_amd import scene, sharding  # noqa: E402
from conftest import scene_params  # noqa: E402


AXIS = 0  # tiling axis of the case being run


def _case(seed, n_target, n_samples):
    xyz, ws = scene.make_scene(seed=seed, n_target=n_target)
    idx = scene.draw_samples(seed, xyz.shape[0], n_samples)
    return xyz, ws, sharding.order_samples_by_x(xyz, idx, AXIS)


def _halo(det):
    q = det.params
    return sharding.tile_halo(q.nn_radius_hands, q.nn_radius_taubin, q.normals_radius)


def _run_tiled(make, xyz, ordered, world, R, seed, halo):
    origin = sharding.cloud_origin(xyz)
    pad = sharding.max_shard(len(ordered), world)
    tabs, keeps, fractions = [], [], []
    for rank in range(world):
        keep, local, base = sharding.tile_points(xyz, ordered, rank, world, halo, AXIS)
        det = make()
        if len(local) == 0:
            recs = np.zeros(0, dtype=HYP)
            pr = np.zeros(0, np.uint8)
        else:
            det.set_grid_origin(origin)
            det.set_cloud(xyz[keep])
            det.compute_normals()
            recs = det.generate_hypotheses(sample_idx=local, slot_base=base, seed=seed)
            pr = det.prune(len(recs)) if len(recs) else np.zeros(0, np.uint8)
        fractions.append(len(keep) / xyz.shape[0])
        tabs.append(sharding.table_from_records(recs, base, len(local), R, pad))
        keeps.append(pr)
        det.close()
    return sharding.compact_table(np.concatenate(tabs)), np.concatenate(keeps), fractions


HYP = None  # record dtype of the binding under test (identical layouts), set by each test


def _check(make, world, seed=5, n_target=9000, n_samples=53, R=8):
    xyz, ws, ordered = _case(seed, n_target, n_samples)
    p = scene_params(ws, num_threads=2)
    full = make(p)
    halo = _halo(full)
    full.set_cloud(xyz)
    full.compute_normals()
    want = full.generate_hypotheses(sample_idx=ordered, slot_base=0, seed=seed)
    want_keep = full.prune(len(want))
    full.close()
    assert len(want) > 10
    got, got_keep, fr = _run_tiled(lambda: make(p), xyz, ordered, world, R, seed, halo)
    assert got.tobytes() == want.tobytes()
    assert (got_keep == want_keep).all()
    return fr


@pytest.mark.parametrize("world,axis", [(2, 0), (3, 0), (5, 0), (3, 1)])
def test_oracle_tiles_equal_unsplit(world, axis):
    global HYP, AXIS
    from oracle import api
    HYP = api.HYP_DTYPE
    AXIS = axis
    try:
        fr = _check(lambda p: api.Oracle(**p), world)
    finally:
        AXIS = 0
    assert min(fr) < 1.0          # at least one tile really dropped points


def test_tile_helpers():
    rng = np.random.default_rng(0)
    xyz = rng.uniform(-1, 1, size=(2000, 3)).astype(np.float32)
    xyz[7] = np.nan
    idx = rng.choice(2000, 40, replace=False)
    idx = idx[idx != 7]
    ordered = sharding.order_samples_by_x(xyz, idx)
    assert (np.diff(xyz[ordered, 0]) >= 0).all() and sorted(ordered) == sorted(idx)
    assert (sharding.cloud_origin(xyz) == np.nanmin(xyz, axis=0)).all()
    seen = []
    for r in range(4):
        keep, local, base = sharding.tile_points(xyz, ordered, r, 4, 0.1)
        assert (np.diff(keep) > 0).all() and 7 not in keep
        assert (keep[local] == ordered[base:base + len(local)]).all()
        seen += list(keep[local])
        x = xyz[keep, 0]
        assert x.min() >= xyz[keep[local], 0].min() - 0.1 - 1e-6
    assert seen == list(ordered)
    assert sharding.longest_axis(xyz * np.float32([1, 3, 2])) == 1
    # more ranks than samples: empty tiles
    keep, local, base = sharding.tile_points(xyz, ordered[:2], 3, 4, 0.1)
    assert len(keep) == 0 and len(local) == 0


def test_origin_above_a_point_is_an_error():
    from oracle import api
    xyz, ws = scene.make_scene(seed=1, n_target=2000)
    o = api.Oracle(**scene_params(ws, num_threads=1))
    o.set_grid_origin(xyz.min(axis=0) + np.float32(0.01))
    with pytest.raises(RuntimeError):
        o.set_cloud(xyz)
    o.set_grid_origin(None)
    o.set_cloud(xyz)


@pytest.mark.gpu
@pytest.mark.parametrize("world,axis", [(2, 0), (4, 0), (4, 1)])
def test_gpu_tiles_equal_unsplit(world, axis):
    global HYP, AXIS
    from agile_grasp2_amd import capi
    HYP = capi.HYP_DTYPE
    AXIS = axis
    try:
        fr = _check(lambda p: capi.Detector(**p), world, seed=9, n_target=60000, n_samples=301)
    finally:
        AXIS = 0
    assert min(fr) < 1.0


@pytest.mark.gpu
def test_gpu_tiles_equal_oracle_tiles():
    """The tile a rank computes on the GPU equals the same tile computed by the oracle."""
    from agile_grasp2_amd import capi
    from oracle import api
    xyz, ws, ordered = _case(4, 20000, 97)
    p = scene_params(ws, num_threads=2)
    d = capi.Detector(**p)
    halo = _halo(d)
    d.close()
    assert abs(halo - 0.11) < 1e-4
    origin = sharding.cloud_origin(xyz)
    keep, local, base = sharding.tile_points(xyz, ordered, 1, 3, halo)
    out = []
    for make in (lambda: capi.Detector(**p), lambda: api.Oracle(**p)):
        d = make()
        d.set_grid_origin(origin)
        d.set_cloud(xyz[keep])
        d.compute_normals()
        out.append(d.generate_hypotheses(sample_idx=local, slot_base=base, seed=2))
        d.close()
    assert len(out[0]) > 5 and out[0].tobytes() == out[1].tobytes()


@pytest.mark.gpu
def test_gpu_origin_above_a_point_is_an_error():
    from agile_grasp2_amd import capi
    xyz, ws = scene.make_scene(seed=1, n_target=2000)
    d = capi.Detector(**scene_params(ws, num_threads=1))
    d.set_grid_origin(xyz.min(axis=0) + np.float32(0.01))
    with pytest.raises(RuntimeError):
        d.set_cloud(xyz)
        d.compute_normals()
    d.set_grid_origin(None)
    d.set_cloud(xyz)
    d.compute_normals()
    d.close()
