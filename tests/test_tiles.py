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


@pytest.mark.gpu
@pytest.mark.parametrize("world,min_inliers", [(1, 0), (3, 0), (1, 1), (3, 1)])
def test_merge_of_the_ranks_selected_lists_on_the_gpu(world, min_inliers):
    """The multi-GPU merge (ag2_export_selected_compact_device on every rank, all-gather,
    ag2_merge_selected_device): the ranks' selected lists concatenated in rank order and the top
    num_selected by score, ties by position -- checked against the same merge done in numpy on the
    exported bytes, and against the unsplit run (same hypotheses; the scores of a tile can differ from
    the unsplit run's in the last bits because ip1's split-K depends on the batch size).
    min_inliers > 0: HandleSearch::findClusters counts inliers over the hands of ALL ranks
    (handle_search.cpp:4-80, grasp_detector.cpp:228-236), so the ranks export their lists before the
    clustering and the merge clusters the gathered list -- checked against the oracle's findClusters on the
    exported bytes, and (one rank) against the clustering ag2_detect does by itself."""
    import ctypes as C
    from agile_grasp2_amd import capi
    from agile_grasp2_amd.weights import make_lenet_weights
    from oracle import api
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    xyz, ws, ordered = _case(7, 12000, 150)
    w = make_lenet_weights(7)
    # threshold = the median score of the unsplit run: about half of the records take part in the merge
    d0 = capi.Detector(**scene_params(ws, min_score_diff=-1e30, num_selected=-1))
    d0.lenet_load(w)
    d0.set_cloud(xyz)
    d0.compute_normals()
    _, all0 = d0.detect(sample_idx=ordered, seed=3, do_prune=False)
    d0.close()
    thr = float(np.median(all0["score"]))
    prm = scene_params(ws, min_score_diff=thr, num_selected=17)
    R, cap = 8, 150 * 8
    per = sharding.compact_bytes(cap)
    dbuf = C.c_void_p()
    assert hip.hipMalloc(C.byref(dbuf), per * world) == 0
    halo = None
    costs = sharding.sample_costs(xyz, ordered, 0.1, AXIS)
    bounds = sharding.balanced_bounds(costs, world)
    assert bounds[0] == 0 and bounds[-1] == len(ordered) and np.all(np.diff(bounds) > 0)
    dets = []
    for rank in range(world):
        d = capi.Detector(**prm)
        d.lenet_load(w)
        halo = halo or _halo(d)
        keep, local, base = sharding.tile_points(xyz, ordered, rank, world, halo, AXIS, bounds)
        d.set_grid_origin(sharding.cloud_origin(xyz))
        d.set_cloud(xyz[keep])
        d.compute_normals()
        d.set_min_inliers(min_inliers)
        # a rank of a multi-GPU run (no local selection), or -- rank 0 -- a caller that also selects locally
        d.detect(sample_idx=local, slot_base=base, seed=3, do_prune=False, want_all=False, local_select=rank == 0)
        d.export_selected_compact_device(dbuf.value + rank * per, per, cap)
        dets.append(d)
    hip.hipDeviceSynchronize()   # (every rank's export -- on its own stream -- before anyone reads the buffer)
    raw = np.zeros(per * world, dtype=np.uint8)
    assert hip.hipMemcpy(raw.ctypes.data_as(C.c_void_p), dbuf, per * world, 2) == 0
    flat, cut = sharding.unpack_compact(raw, world, cap, capi.HYP_DTYPE)
    assert not cut and len(flat) > 17
    pool = flat
    if min_inliers > 0:
        pool = api.Oracle(**prm).find_clusters(flat, min_inliers)
        assert 0 < len(pool) <= len(flat)
    order = sorted(range(len(pool)), key=lambda i: (-pool["score"][i], i))[:17]
    want = pool[order]
    got, n_total = dets[0].merge_selected_device(dbuf.value, world, cap)
    assert n_total == len(flat) and got.tobytes() == want.tobytes()
    # a rank's list longer than the exchange capacity voids the merge: an error, not a silent cut
    small = max(1, int(raw[:4].view(np.uint32)[0]) // 2)
    per_s = sharding.compact_bytes(small)
    dets[0].export_selected_compact_device(dbuf.value, per_s, small)
    with pytest.raises(RuntimeError, match="exchange capacity"):
        dets[0].merge_selected_device(dbuf.value, 1, small)
    # the unsplit run selects from the same hypotheses
    d0 = capi.Detector(**prm)
    d0.lenet_load(w)
    d0.set_cloud(xyz)
    d0.compute_normals()
    d0.set_min_inliers(min_inliers)
    sel0, all0 = d0.detect(sample_idx=ordered, seed=3, do_prune=False)
    thr_margin = np.abs(all0["score"] - thr).min()
    if thr_margin > 1e-2:
        keys = sorted(zip(flat["sample_slot"], flat["orientation"]))
        assert keys == sorted((s, o) for s, o, sc in zip(all0["sample_slot"], all0["orientation"], all0["score"]) if sc >= thr)
    if world == 1:
        assert got.tobytes() == sel0.tobytes()
    for d in dets + [d0]:
        d.close()
    hip.hipFree(dbuf)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ties", "long", "not_a_float"])
def test_device_top_k_order(case):
    """The device top-k (merge, frame mode, ag2_detect's one-round-trip form) orders by score descending,
    ties by position (grasp_detector.cpp:239-252 with a stable tie rule).  Crafted lists: equal scores,
    +0.0 against -0.0, negative scores and infinities ("ties"); more records than the kernel's key stage
    holds ("long": the general loop); a score that is not exactly a float ("not_a_float": the general
    loop again) -- against the same order taken in numpy."""
    import ctypes as C
    from agile_grasp2_amd import capi
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    rng = np.random.default_rng(11)
    world = 3
    n_rank = {"ties": (200, 0, 317), "long": (2500, 2100, 1900), "not_a_float": (90, 40, 70)}[case]
    cap = max(n_rank)
    lists = []
    for r, n in enumerate(n_rank):
        rec = np.zeros(n, dtype=capi.HYP_DTYPE)
        sc = rng.choice(np.float32([-3.5, -0.0, 0.0, 1.25, 1.25, 7.0, -1e30, 2.0 ** -130]), size=n).astype(np.float64)
        if case == "long":
            sc = rng.integers(-50, 50, size=n).astype(np.float64)       # many ties
        if case == "ties" and n:
            sc[:4] = [np.inf, -np.inf, 0.0, -0.0]
        if case == "not_a_float" and r == 1:
            sc[7] = 0.1                                                  # a double no float equals
        rec["score"] = sc
        rec["sample_slot"] = 1000 * r + np.arange(n)
        rec["n_points"] = 1 + np.arange(n)
        lists.append(rec)
    per = sharding.compact_bytes(cap)
    raw = np.concatenate([sharding.pack_compact(rec, cap) for rec in lists])
    flat = np.concatenate(lists)
    dbuf = C.c_void_p()
    assert hip.hipMalloc(C.byref(dbuf), per * world) == 0
    assert hip.hipMemcpy(dbuf, raw.ctypes.data_as(C.c_void_p), per * world, 1) == 0
    ws = [0.0, 1.0, -1.0, 1.0, -1.0, 1.0]
    for k in (1, 29, -1):
        d = capi.Detector(**scene_params(ws, num_selected=k))
        got, n_total = d.merge_selected_device(dbuf.value, world, cap)
        order = sorted(range(len(flat)), key=lambda i: (-flat["score"][i], i))
        want = flat[order if k < 0 else order[:k]]
        assert n_total == len(flat)
        assert got.tobytes() == want.tobytes(), (case, k)
        d.close()
    hip.hipFree(dbuf)


def test_cost_balanced_bounds():
    rng = np.random.default_rng(0)
    costs = rng.integers(1, 100, size=1000).astype(np.float64)
    for world in (1, 2, 3, 8):
        b = sharding.balanced_bounds(costs, world)
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b) > 0)
        sums = [costs[b[r]:b[r + 1]].sum() for r in range(world)]
        assert max(sums) <= costs.sum() / world + 100
    assert list(sharding.balanced_bounds(np.ones(3), 8))[:4] == [0, 1, 2, 3]
    assert list(sharding.balanced_bounds(np.zeros(0), 4)) == [0, 0, 0, 0, 0]


@pytest.mark.gpu
def test_a_rank_detects_in_one_trip_and_a_shape_that_does_not_hold_makes_every_rank_repeat():
    """A rank of a multi-GPU job (ag2_detect without a result buffer) waits for NOTHING from its second call on:
    the tail is launched at the shapes its previous call left, the exported header says whether they held
    ({count, cap, status, images scored}), and the merge -- which every rank runs on the same gathered bytes --
    answers AG2_ERR_RETRY when some rank's did not, so that all repeat the step (round 4).
    Two ranks on the one GPU: (1) repeated steps return the bytes of the first, step-by-step one, the second and
    third in one trip (counters); (2) rank 1 then meets a cloud with far more hypotheses than its shapes hold:
    its header carries status 1 and no records, the merge raises RetryStep on EVERY rank, and the repeated step
    gives the bytes two fresh contexts give."""
    import ctypes as C
    from agile_grasp2_amd import capi
    from agile_grasp2_amd.weights import make_lenet_weights
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    w = make_lenet_weights(7)
    bare, ws1 = scene.make_scene(seed=21, n_target=60000, kind="plane")
    busy, ws2 = scene.make_scene(seed=22, n_target=60000, kind="tabletop")
    wsu = [min(ws1[0], ws2[0]), max(ws1[1], ws2[1]), min(ws1[2], ws2[2]), max(ws1[3], ws2[3]),
           min(ws1[4], ws2[4]), max(ws1[5], ws2[5])]
    prm = scene_params(wsu, num_selected=20, min_score_diff=-1e30)
    s = 2500
    lo, hi = bare.min(axis=0), bare.max(axis=0)
    inner = np.flatnonzero((bare[:, 0] > lo[0] + 0.2) & (bare[:, 0] < hi[0] - 0.2) &
                           (bare[:, 1] > lo[1] + 0.2) & (bare[:, 1] < hi[1] - 0.2)).astype(np.int32)
    i_bare = inner[scene.draw_samples(1, len(inner), s)]     # no hand finds anything to close around
    i_busy = scene.draw_samples(2, busy.shape[0], s)
    cap = s * 8
    per = sharding.compact_bytes(cap)
    dbuf = C.c_void_p()
    assert hip.hipMalloc(C.byref(dbuf), per * 2) == 0

    def make():
        d = capi.Detector(**prm)
        d.lenet_load(w)
        return d

    def step(dets, clouds, idxs):
        """one step of both ranks: detect (no local selection), export, 'all-gather' (adjacent buffers), merge on both"""
        for r, d in enumerate(dets):
            d.set_cloud(clouds[r])
            d.compute_normals()
            d.detect(sample_idx=idxs[r], slot_base=r * s, seed=4, do_prune=False, want_all=False, local_select=False)
            d.export_selected_compact_device(dbuf.value + r * per, per, cap)
        assert hip.hipDeviceSynchronize() == 0   # (what the all-gather orders: every rank's export before any merge)
        return [d.merge_selected_device(dbuf.value, 2, cap) for d in dets]

    def headers():
        raw = np.zeros(per * 2, dtype=np.uint8)
        assert hip.hipMemcpy(raw.ctypes.data_as(C.c_void_p), dbuf, per * 2, 2) == 0
        return raw.reshape(2, per)[:, :16].copy().view(np.uint32)

    dets = [make(), make()]
    # (1) rank 0 on the busy cloud, rank 1 on the bare table, three times
    first = step(dets, [busy, bare], [i_busy, i_bare])
    n0 = int(dets[0].counters().n_scored)
    assert n0 > 600 and len(first[0][0]) == 20 and first[0][0].tobytes() == first[1][0].tobytes()
    assert [int(d.counters().detect_one_trip) for d in dets] == [0, 0]
    for rep in range(2):
        again = step(dets, [busy, bare], [i_busy, i_bare])
        for r in range(2):
            assert again[r][0].tobytes() == first[0][0].tobytes() and again[r][1] == first[0][1], (rep, r)
        h = headers()
        assert h[0, 2] == 0 and h[1, 2] == 0 and h[0, 3] == n0 and h[1, 3] == 0 and h[0, 0] == n0 and h[1, 0] == 0
        assert int(dets[0].counters().n_scored) == n0 and int(dets[1].counters().n_scored) == 0
        assert dets[0].times().total_ms > 0
    assert [int(d.counters().detect_one_trip) for d in dets] == [2, 2]
    assert [int(d.counters().detect_redone) for d in dets] == [0, 0]
    # (2) rank 1 now gets the busy cloud too: 256 images of room, hundreds of hypotheses
    for r, d in enumerate(dets):
        d.set_cloud(busy)
        d.compute_normals()
        d.detect(sample_idx=i_busy, slot_base=r * s, seed=4, do_prune=False, want_all=False, local_select=False)
        d.export_selected_compact_device(dbuf.value + r * per, per, cap)
    assert hip.hipDeviceSynchronize() == 0
    for d in dets:                       # every rank takes the same decision from the same bytes
        with pytest.raises(capi.RetryStep, match="every rank repeats"):
            d.merge_selected_device(dbuf.value, 2, cap)
    h = headers()                        # (read after the merges: they waited for the ranks' streams)
    assert h[0, 2] == 0 and h[0, 0] == n0 and h[1, 2] == 1 and h[1, 0] == 0 and h[1, 3] > 256
    assert int(dets[1].counters().detect_redone) == 1 and int(dets[0].counters().detect_redone) == 0
    redo = step(dets, [busy, busy], [i_busy, i_busy])
    fresh = [make(), make()]
    want = step(fresh, [busy, busy], [i_busy, i_busy])
    for r in range(2):
        assert redo[r][0].tobytes() == want[0][0].tobytes() and redo[r][1] == want[0][1] > n0
    settled = step(dets, [busy, busy], [i_busy, i_busy])           # and the next one is one trip again
    assert settled[0][0].tobytes() == want[0][0].tobytes()
    assert int(dets[1].counters().detect_one_trip) == 3 and int(dets[0].counters().detect_one_trip) == 4
    for d in dets + fresh:
        d.close()
    hip.hipFree(dbuf)


def test_cpp_tile_plan_equals_sharding_py():
    """GraspDetector's spatial tiling (Params::tiling = spatial, agile_grasp2_amd/host/ag2_host.cpp) is a port of
    sharding.py: the sample order along the longest axis, the cost-balanced bounds and the tiles' sizes must be the
    same for any cloud -- held against each other on the CPU (ag2host_tile_plan touches no GPU), over random clouds
    with non-finite points, duplicate coordinates, more ranks than samples."""
    import ctypes as C
    import subprocess
    host_dir = os.path.join(ROOT, "agile_grasp2_amd", "host")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "agile_grasp2_amd", "csrc"), "-s", "-j", "8"])
    subprocess.check_call(["make", "-C", host_dir, "-s"])
    lib = C.CDLL(os.path.join(host_dir, "libag2host.so"))
    lib.ag2host_tile_plan.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_double,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(3)
    for case in range(12):
        n = int(rng.integers(50, 4000))
        scale = np.float32([1.0, 1.0, 1.0])
        scale[case % 3] = 3.0                       # the longest axis varies
        xyz = (rng.uniform(-0.3, 0.3, size=(n, 3)).astype(np.float32) * scale).astype(np.float32)
        if case % 2:
            xyz[:, case % 3] = np.round(xyz[:, case % 3], 2)     # many equal coordinates along the tiling axis: ties
        bad = rng.choice(n, 3, replace=False)
        xyz[bad[0]] = np.nan
        xyz[bad[1], 1] = np.inf
        s = int(rng.integers(1, 60))
        good = np.setdiff1d(np.arange(n), bad)
        idx = np.sort(rng.choice(good, s, replace=False)).astype(np.int32)
        for world in (1, 2, 3, 8, 64):
            axis = sharding.longest_axis(xyz)
            ordered = sharding.order_samples_by_x(xyz, idx, axis)
            bounds = sharding.balanced_bounds(sharding.sample_costs(xyz, ordered, 0.1, axis), world)
            halo = sharding.tile_halo(0.1, 0.01, 0.01)
            tiles = [len(sharding.tile_points(xyz, ordered, r, world, halo, axis, bounds)[0]) for r in range(world)]
            o = np.zeros(s, np.int32)
            b = np.zeros(world + 1, np.int64)
            t = np.zeros(world, np.int64)
            ax = C.c_int32(-1)
            rc = lib.ag2host_tile_plan(xyz.ctypes.data, n, idx.ctypes.data, s, world, 0.1, halo, o.ctypes.data,
                                       b.ctypes.data, t.ctypes.data, C.byref(ax))
            assert rc == 0 and ax.value == axis, (case, world)
            assert np.array_equal(o, ordered), (case, world)
            assert np.array_equal(b, bounds), (case, world, b, bounds)
            assert list(t) == tiles, (case, world)
