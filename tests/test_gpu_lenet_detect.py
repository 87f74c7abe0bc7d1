"""GPU parity, K5 + K6 + the fused detect entry: LeNet logits and the selected grasps, HIP (through
the C-ABI) vs the oracle (fp32 CPU restatement of the Caffe layers) and vs torch-CPU.

LeNet bar: floating point, different summation order (MFMA k-ordered fma chain vs the oracle's
separate mul/add, K permuted for the LDS layout): |diff| <= 1e-4 * max|logit| + 1e-3, the same
tolerance the oracle itself is pinned to torch with.  Selection: the set and order of selected
hypotheses must be identical; scores within the same tolerance.
"""
import os

import numpy as np
import pytest

import np_reference as ref
from conftest import scene_params
from agile_grasp2_amd import scene
from agile_grasp2_amd.weights import make_lenet_weights

pytestmark = pytest.mark.gpu


def random_images(rng, n, density=0.1):
    m = rng.uniform(0, 1, size=(n, 60, 60, 3)) < density
    return (m * rng.integers(0, 256, size=(n, 60, 60, 3))).astype(np.uint8)


def tol(want):
    return 1e-4 * np.abs(want).max() + 1e-3


@pytest.mark.parametrize("n", [1, 5, 64, 130])
def test_lenet_matches_oracle_and_torch(n):
    from agile_grasp2_amd import capi
    from oracle import api
    w = make_lenet_weights(11)
    rng = np.random.default_rng(n)
    imgs = random_images(rng, n)
    if n >= 5:
        imgs[1] = 0
        imgs[2] = 255
    d = capi.Detector()
    d.lenet_load(w)
    got = d.lenet_forward(imgs)
    o = api.Oracle(num_threads=8)
    o.lenet_load(w)
    want = o.lenet_forward(imgs)
    wt = ref.lenet_torch(w, imgs)
    assert np.abs(got - want).max() <= tol(want)
    assert np.abs(got - wt).max() <= tol(wt)
    d.close()


def test_lenet_structured_weights_catch_layout_bugs():
    """Asymmetric, position-dependent weights: a transposed tile, a swapped channel pair or a
    mis-ordered ip1 column would change the answer by O(1)."""
    from agile_grasp2_amd import capi
    w = make_lenet_weights(3)
    rng = np.random.default_rng(0)
    w["conv1_w"] = (np.arange(20 * 3 * 25).reshape(20, 3, 5, 5) % 17 - 8).astype(np.float32) * 1e-3
    w["conv2_w"] = (np.arange(50 * 20 * 25).reshape(50, 20, 5, 5) % 23 - 11).astype(np.float32) * 1e-3
    w["ip1_w"] = ((np.arange(500 * 7200).reshape(500, 7200) % 31) - 15).astype(np.float32) * 1e-4
    w["ip2_w"] = np.stack([np.linspace(-1, 1, 500), np.linspace(1, -0.5, 500)]).astype(np.float32)
    imgs = random_images(rng, 9, density=0.5)
    d = capi.Detector()
    d.lenet_load(w)
    got = d.lenet_forward(imgs)
    wt = ref.lenet_torch(w, imgs)
    assert np.abs(got - wt).max() <= tol(wt)
    d.close()


def _detect_pair(xyz, ws, idx, seed, do_prune, **kw):
    from agile_grasp2_amd import capi
    from oracle import api
    prm = scene_params(ws, **kw)
    w = make_lenet_weights(7)
    o = api.Oracle(**dict(prm, num_threads=8))
    d = capi.Detector(**prm)
    for x in (o, d):
        x.set_cloud(xyz)
        x.compute_normals()
        x.lenet_load(w)
    gs, ga = d.detect(sample_idx=idx, seed=seed, do_prune=do_prune)
    ws_, wa = o.detect(sample_idx=idx, seed=seed, do_prune=do_prune)
    return d, o, gs, ga, ws_, wa


def _check_scored(ga, wa):
    assert len(ga) == len(wa)
    for f in ("sample_slot", "orientation", "half_antipodal", "full_antipodal", "n_points"):
        assert np.array_equal(ga[f], wa[f]), f
    for f in ("axis", "approach", "binormal", "surface", "bottom", "top", "width"):
        assert np.array_equal(ga[f], wa[f]), f
    if len(wa):
        assert np.abs(ga["score"] - wa["score"]).max() <= 2 * tol(wa["score"])


def test_detect_small_scene(small_scene):
    xyz, ws, idx = small_scene
    med = None
    d, o, gs, ga, ws_, wa = _detect_pair(xyz, ws, idx, 5, True, min_score_diff=-1e30, num_selected=1000)
    _check_scored(ga, wa)
    assert len(wa) > 5
    # with everything selected the order is by score: identical unless two scores are closer than
    # the float tolerance
    assert len(gs) == len(ws_) == len(wa)
    key_g = list(zip(gs["sample_slot"], gs["orientation"]))
    key_w = list(zip(ws_["sample_slot"], ws_["orientation"]))
    assert sorted(key_g) == sorted(key_w)
    # the ORDER is always checked: the two lists may differ only by swaps of records whose oracle scores lie
    # within 2 tol of each other (never degraded to a set compare)
    from agile_grasp2_amd.selection_check import check_selection
    check_selection(gs, wa, -1e30, 1000, tol(wa["score"]), tag="small scene")
    score_of = {k: float(s) for k, s in zip(zip(wa["sample_slot"], wa["orientation"]), wa["score"])}
    for a, b in zip(key_g, key_w):
        assert a == b or abs(score_of[a] - score_of[b]) <= 2 * tol(wa["score"]), (a, b)
    assert gs["full_antipodal"].all()
    d.close()


def test_detect_threshold_and_topk(small_scene):
    xyz, ws, idx = small_scene
    d, o, gs, ga, ws_, wa = _detect_pair(xyz, ws, idx, 6, False, min_score_diff=-1e30, num_selected=1000)
    thr = float(np.median(wa["score"]))
    margin = np.abs(wa["score"] - thr).min()
    d.close()
    d, o, gs, ga, ws_, wa = _detect_pair(xyz, ws, idx, 6, False, min_score_diff=thr, num_selected=10)
    _check_scored(ga, wa)
    assert len(ws_) == min(10, int((wa["score"] >= thr).sum()))
    from agile_grasp2_amd.selection_check import check_selection
    chk = check_selection(gs, wa, thr, 10, tol(wa["score"]), max_uncertain=4, tag="threshold + top-10")
    if margin > 4 * tol(wa["score"]):   # (the threshold is the median: one record sits on it by construction)
        assert len(gs) == len(ws_)
    c = d.counters()
    assert c.n_scored == len(wa) and c.n_selected == len(gs)
    t = d.times()
    assert t.total_ms > 0 and t.lenet_conv_ms > 0 and t.sweep_ms > 0
    d.close()


def test_detect_on_the_null_stream_with_growing_lists(small_scene):
    """ag2_set_stream(NULL) = the HIP default stream (what torch's default stream handle is).  The
    page-locked staging area is re-allocated while the lists grow; every such re-allocation has to
    wait for the stream although its handle is NULL."""
    from agile_grasp2_amd import capi
    from oracle import api
    xyz, ws, idx = small_scene
    prm = scene_params(ws, min_score_diff=-1e30, num_selected=100000)
    d = capi.Detector(**prm)
    d.set_stream(0)
    o = api.Oracle(**dict(prm, num_threads=4))
    w = make_lenet_weights(3)
    for x in (d, o):
        x.set_cloud(xyz)
        x.compute_normals()
        x.lenet_load(w)
    big = scene.draw_samples(11, xyz.shape[0], 3000)
    for k, samples in enumerate((idx[:10], idx, big[:700], big)):
        gs, ga = d.detect(sample_idx=samples, seed=k, do_prune=False)
        ws_, wa = o.detect(sample_idx=samples, seed=k, do_prune=False)
        _check_scored(ga, wa)
        assert len(gs) == len(ws_)
    d.close()


def test_one_round_trip_detect_equals_the_step_by_step_form(small_scene):
    """From its second call on a context, ag2_detect launches its tail at the shapes the previous call
    left and picks the top-k on the device (one host round trip); asking for all scored records keeps
    the step-by-step form.  Same bytes either way, also when the shapes do not hold (a busier cloud:
    the call notices at its end and runs again step by step)."""
    from agile_grasp2_amd import capi
    xyz, ws, idx = small_scene
    w = make_lenet_weights(5)
    for nsel, thr in ((7, -1e30), (-1, -1e30), (30, None)):
        prm = scene_params(ws, num_selected=nsel, **({} if thr is None else {"min_score_diff": thr}))
        if thr is None:
            prm = dict(prm, min_score_diff=0.0)
        d = capi.Detector(**prm)
        d.set_cloud(xyz)
        d.compute_normals()
        d.lenet_load(w)
        ref_sel, ref_all = d.detect(sample_idx=idx, seed=3, do_prune=True)             # step by step
        ref_cnt = d.counters()
        for rep in range(3):
            sel, n_scored = d.detect(sample_idx=idx, seed=3, do_prune=True, want_all=False)
            assert sel.tobytes() == ref_sel.tobytes(), (nsel, rep)
            assert n_scored == len(ref_all)
            c = d.counters()
            for f in ("n_frames", "n_hypotheses", "sum_kcrop", "sum_p", "n_scored", "n_selected", "n_pruned"):
                assert getattr(c, f) == getattr(ref_cnt, f), f
            assert d.times().total_ms > 0
        # another seed / prune setting through the same context
        a_sel, a_all = d.detect(sample_idx=idx, seed=4, do_prune=False)
        b_sel, b_n = d.detect(sample_idx=idx, seed=4, do_prune=False, want_all=False)
        c_sel, c_n = d.detect(sample_idx=idx, seed=4, do_prune=False, want_all=False)
        assert b_sel.tobytes() == a_sel.tobytes() == c_sel.tobytes() and b_n == c_n == len(a_all)
        d.close()
    # shapes that do not hold: learn on a bare table (no hypotheses: capacity for 256 images), then a
    # cluttered one with the same number of samples
    bare, ws1 = scene.make_scene(seed=21, n_target=60000, kind="plane")
    busy, ws2 = scene.make_scene(seed=22, n_target=60000, kind="tabletop")
    wsu = [min(ws1[0], ws2[0]), max(ws1[1], ws2[1]), min(ws1[2], ws2[2]), max(ws1[3], ws2[3]),
           min(ws1[4], ws2[4]), max(ws1[5], ws2[5])]
    prm = scene_params(wsu, num_selected=20, min_score_diff=-1e30)
    d, e = capi.Detector(**prm), capi.Detector(**prm)
    for x in (d, e):
        x.lenet_load(w)
    s = 2500
    # samples of the bare table well away from its border: no hand finds anything to close around
    lo, hi = bare.min(axis=0), bare.max(axis=0)
    inner = np.flatnonzero((bare[:, 0] > lo[0] + 0.2) & (bare[:, 0] < hi[0] - 0.2) &
                           (bare[:, 1] > lo[1] + 0.2) & (bare[:, 1] < hi[1] - 0.2)).astype(np.int32)
    assert len(inner) > s
    i1 = inner[scene.draw_samples(1, len(inner), s)]
    i2 = scene.draw_samples(2, busy.shape[0], s)
    d.set_cloud(bare)
    d.compute_normals()
    for _ in range(2):
        _, n_bare = d.detect(sample_idx=i1, seed=1, do_prune=False, want_all=False)
        assert n_bare == 0
    d.set_cloud(busy)
    d.compute_normals()
    got, n_got = d.detect(sample_idx=i2, seed=2, do_prune=False, want_all=False)   # must fall back
    again, n_again = d.detect(sample_idx=i2, seed=2, do_prune=False, want_all=False)
    e.set_cloud(busy)
    e.compute_normals()
    want, want_all = e.detect(sample_idx=i2, seed=2, do_prune=False)
    assert len(want_all) > 300 and n_got == n_again == len(want_all)
    assert got.tobytes() == want.tobytes() == again.tobytes()
    # the counters say which form served the calls: bare #2 in one trip; busy #1 launched at the bare
    # table's shapes (256 images), found > 300 and ran again; busy #2 in one trip at the new shapes
    cd, ce = d.counters(), e.counters()
    if not os.environ.get("AG2_DETECT_STEPWISE"):   # (the A/B switch keeps every call step by step)
        assert (cd.detect_one_trip, cd.detect_redone) == (2, 1), (cd.detect_one_trip, cd.detect_redone)
    assert (ce.detect_one_trip, ce.detect_redone) == (0, 0)
    d.close()
    e.close()


@pytest.mark.parametrize("num_selected", [5, 64, 1000])
def test_device_top_k_breaks_ties_by_list_position(small_scene, num_selected):
    """All-zero weights: every image scores exactly 0.0, so the order of the selection is nothing but the
    tie rule -- list position, ascending (the step-by-step form's host sort, grasp_detector.cpp:239-252) --
    in the one-round-trip form (k_topk on the device: ranks from 64-bit keys) as in the step-by-step one.  A second weight set makes every score one of two values (ip2 bias only)."""
    from agile_grasp2_amd import capi
    xyz, ws, idx = small_scene
    w0 = {k: np.zeros_like(v) for k, v in make_lenet_weights(5).items()}
    for bias in (0.0, 0.5):
        w = {k: v.copy() for k, v in w0.items()}
        w["ip2_b"][1] = bias                                  # score = b[1] - b[0] for every image
        d = capi.Detector(**scene_params(ws, num_selected=num_selected, min_score_diff=0.0))
        d.set_cloud(xyz)
        d.compute_normals()
        d.lenet_load(w)
        ref_sel, ref_all = d.detect(sample_idx=idx, seed=3, do_prune=False)              # step by step
        assert len(ref_all) > 70 and np.all(ref_all["score"] == bias)
        want = ref_all[:min(num_selected, len(ref_all))]                                 # list order
        assert [(int(r["sample_slot"]), int(r["orientation"])) for r in ref_sel] == \
            [(int(r["sample_slot"]), int(r["orientation"])) for r in want]
        for rep in range(2):
            sel, n_scored = d.detect(sample_idx=idx, seed=3, do_prune=False, want_all=False)  # one round trip
            assert n_scored == len(ref_all) and sel.tobytes() == ref_sel.tobytes(), (bias, rep)
        assert d.counters().detect_one_trip == (0 if os.environ.get("AG2_DETECT_STEPWISE") else 2)
        d.close()


def test_detect_with_more_than_8192_scored_images():
    """Beyond 8 192 scored images the threshold / compaction runs as flags + chained scan + gather instead of
    the one-workgroup kernel, and beyond 4 096 selected records the device top-k leaves its key stage for the
    general compare loop (k_score_flags, k_gather_selected, k_topk / host sort): the selection must be the
    scored list filtered by the threshold and stably sorted by score (grasp_detector.cpp:200-252)."""
    from agile_grasp2_amd import capi
    xyz, ws = scene.make_scene(seed=5, n_target=250000, kind="tabletop", voxel=None)
    idx = scene.draw_samples(5, xyz.shape[0], 6000)
    w = make_lenet_weights(5)
    for nsel, thr in ((200, 0.0), (6000, -1e30)):
        d = capi.Detector(**scene_params(ws, num_orientations=16, min_score_diff=thr, num_selected=nsel))
        d.set_cloud(xyz)
        d.compute_normals()
        d.lenet_load(w)
        sel, allh = d.detect(sample_idx=idx, seed=1, do_prune=False)
        assert len(allh) > 8192, len(allh)
        passing = allh[allh["score"] >= thr]
        assert len(passing) > (4096 if thr < 0 else 1000)
        want = passing[np.argsort(-passing["score"], kind="stable")][:nsel].copy()
        want["full_antipodal"] = 1                      # grasp_detector.cpp:205
        assert len(sel) == len(want) and sel.tobytes() == want.tobytes(), nsel
        # the same without the list of all scored records (the device picks the top-k where it can)
        sel2, n2 = d.detect(sample_idx=idx, seed=1, do_prune=False, want_all=False)
        assert n2 == len(allh) and sel2.tobytes() == sel.tobytes(), nsel
        d.close()


@pytest.mark.parametrize("min_inliers", [1, 3])
def test_one_round_trip_detect_with_grasp_clusters(small_scene, min_inliers):
    """min_inliers > 0 (the launch files' default is 5): HandleSearch::findClusters between the threshold and
    the top-k (grasp_detector.cpp:228-236) runs inside the one-round-trip form as well -- the bytes of the
    step-by-step form, whose clustered selection has its own tests against the oracle (tests/test_clusters.py)."""
    from agile_grasp2_amd import capi
    xyz, ws, idx = small_scene
    d = capi.Detector(**scene_params(ws, num_selected=25, min_score_diff=-1e30))
    d.set_cloud(xyz)
    d.compute_normals()
    d.lenet_load(make_lenet_weights(5))
    d.set_min_inliers(min_inliers)
    ref_sel, ref_all = d.detect(sample_idx=idx, seed=3, do_prune=False)                # step by step
    assert len(ref_all) > 60 and 0 < len(ref_sel) <= 25
    for rep in range(3):
        sel, n_scored = d.detect(sample_idx=idx, seed=3, do_prune=False, want_all=False)
        assert n_scored == len(ref_all) and sel.tobytes() == ref_sel.tobytes(), rep
    if not os.environ.get("AG2_DETECT_STEPWISE"):
        assert d.counters().detect_one_trip == 3
    d.close()


def test_one_round_trip_detect_when_the_long_list_stage_was_left_out():
    """After runs that queued no sample for the sweep's long-list stage the one-round-trip detect leaves that
    launch out; a cloud whose neighbourhoods then do need it (dense, un-voxelised) is noticed at the end of the
    call and repeated step by step: same bytes as a fresh context, and the next call launches the stage again."""
    from agile_grasp2_amd import capi
    sparse, ws1 = scene.make_scene(seed=31, n_target=40000, kind="tabletop")
    dense, ws2 = scene.make_scene(seed=4, n_target=120000, kind="objects", voxel=None)
    wsu = [min(ws1[0], ws2[0]), max(ws1[1], ws2[1]), min(ws1[2], ws2[2]), max(ws1[3], ws2[3]),
           min(ws1[4], ws2[4]), max(ws1[5], ws2[5])]
    prm = scene_params(wsu, num_selected=20, min_score_diff=-1e30)
    w = make_lenet_weights(5)
    d, e = capi.Detector(**prm), capi.Detector(**prm)
    for x in (d, e):
        x.lenet_load(w)
    s = 60
    i1 = scene.draw_samples(1, sparse.shape[0], s)
    i2 = scene.draw_samples(4, dense.shape[0], s)
    d.set_cloud(sparse)
    d.compute_normals()
    for _ in range(4):
        d.detect(sample_idx=i1, seed=1, do_prune=False, want_all=False)
    assert d.counters().n_overflow_samples == 0
    d.set_cloud(dense)
    d.compute_normals()
    got, n_got = d.detect(sample_idx=i2, seed=2, do_prune=False, want_all=False)
    assert d.counters().n_overflow_samples > 0
    again, n_again = d.detect(sample_idx=i2, seed=2, do_prune=False, want_all=False)
    e.set_cloud(dense)
    e.compute_normals()
    want, want_all = e.detect(sample_idx=i2, seed=2, do_prune=False)
    assert n_got == n_again == len(want_all) and len(want_all) > 20
    assert got.tobytes() == want.tobytes() == again.tobytes()
    if not os.environ.get("AG2_DETECT_STEPWISE"):
        assert d.counters().detect_redone >= 1
    d.close()
    e.close()


def test_banded_convolutions_equal_the_whole_image_kernel_bit_for_bit(monkeypatch):
    """k_lenet_conv_x3b (default: a third of an image per workgroup, two workgroups per CU) runs every
    output through the same chain of MFMAs in the same k order as k_lenet_conv_x3 (AG2_LENET_WHOLE=1:
    one workgroup per image)."""
    from agile_grasp2_amd import capi
    rng = np.random.default_rng(5)
    imgs = rng.integers(0, 256, size=(1000, 60, 60, 3), dtype=np.uint8)
    imgs[::7] = 0
    imgs[3::11, :, :, 1] = 255
    w = make_lenet_weights(11)
    out = {}
    for mode in ("bands", "whole"):
        monkeypatch.delenv("AG2_LENET_WHOLE", raising=False)
        if mode == "whole":
            monkeypatch.setenv("AG2_LENET_WHOLE", "1")
        d = capi.Detector()
        d.lenet_load(w)
        out[mode] = [d.lenet_forward(imgs[:n]) for n in (1, 2, 85, 86, 170, 171, 1000)]
        d.close()
    for a, b in zip(out["bands"], out["whole"]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_detect_no_hypotheses():
    """A bare plane yields no hand placements: every stage must cope with zero work."""
    from agile_grasp2_amd import capi
    xyz, ws = scene.make_scene(seed=2, n_target=4000, kind="plane")
    d = capi.Detector(**scene_params(ws))
    d.set_cloud(xyz)
    d.compute_normals()
    d.lenet_load(make_lenet_weights(1))
    idx = scene.draw_samples(1, xyz.shape[0], 50)
    sel, allh = d.detect(sample_idx=idx, seed=1)
    assert len(sel) == 0 and len(allh) == 0
    sel, allh = d.detect(sample_idx=idx[:0], seed=1)
    assert len(sel) == 0
    d.close()


def test_missing_state_errors():
    from agile_grasp2_amd import capi
    d = capi.Detector()
    with pytest.raises(RuntimeError):
        d.compute_normals()                      # no cloud
    d.set_cloud(np.random.default_rng(0).uniform(0, 0.1, size=(100, 3)).astype(np.float32))
    with pytest.raises(RuntimeError):
        d.generate_hypotheses(sample_idx=np.arange(5, dtype=np.int32))  # no normals
    d.compute_normals()
    with pytest.raises(RuntimeError):
        d.detect(sample_idx=np.arange(5, dtype=np.int32))               # no weights
    d.close()


def test_split_bf16_convolutions_are_as_accurate_as_the_fp32_mfma_path(monkeypatch):
    """k_lenet_conv_x3 writes every fp32 operand as three exact bf16 terms and multiplies on the bf16
    matrix cores (conv1: exact products; conv2: terms below 2^-23 |x w| dropped).  Its deviation from
    the oracle (sequential fp32) must be of the size of the f32-input MFMA kernel's own deviation --
    both only re-order fp32 additions -- and far inside the tolerance the other tests use."""
    from agile_grasp2_amd import capi
    from oracle import api
    rng = np.random.default_rng(3)
    imgs = random_images(rng, 300)
    w = make_lenet_weights(9)
    o = api.Oracle()
    o.lenet_load(w)
    want = o.lenet_forward(imgs).astype(np.float64)
    err = {}
    for mode in ("x3", "f32"):
        if mode == "f32":
            monkeypatch.setenv("AG2_LENET_F32", "1")
        else:
            monkeypatch.delenv("AG2_LENET_F32", raising=False)
        d = capi.Detector()
        d.lenet_load(w)  # the environment is read when the weights are packed
        err[mode] = np.abs(d.lenet_forward(imgs).astype(np.float64) - want).max()
        d.close()
    scale = np.abs(want).max()
    assert err["f32"] <= 2e-5 * scale and err["x3"] <= 2e-5 * scale, (err, scale)
    assert err["x3"] <= 4.0 * err["f32"] + 1e-6 * scale, (err, scale)


def test_stage_timing_levels_do_not_change_results(small_scene):
    """ag2_set_stage_timing only removes HIP events: same records, and the stages without events
    report 0 ms."""
    from agile_grasp2_amd import capi
    from agile_grasp2_amd.weights import make_lenet_weights
    xyz, ws, idx = small_scene
    d = capi.Detector(**scene_params(ws, min_score_diff=-1e9))
    d.lenet_load(make_lenet_weights(3))
    outs = []
    for level in (2, 1, 0, 2):
        d.set_stage_timing(level)
        d.set_cloud(xyz)
        d.compute_normals()
        sel, allh = d.detect(sample_idx=idx, seed=5)
        outs.append((sel.tobytes(), allh.tobytes()))
        t = d.times()
        c = d.counters()
        assert c.sum_k1 > 0
        if level == 2:
            assert t.sweep_ms > 0 and t.lenet_conv_ms > 0 and t.normals_ms > 0
        elif level == 1:
            assert t.sweep_ms > 0 and t.lenet_conv_ms == 0 and t.normals_ms == 0
        else:
            assert t.sweep_ms == 0 and t.total_ms == 0
    assert all(o == outs[0] for o in outs[1:])
    with pytest.raises(RuntimeError):
        d.set_stage_timing(3)
    d.close()
