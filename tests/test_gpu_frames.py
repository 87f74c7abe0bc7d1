"""BASELINE.json configuration 5: a stream of clouds through ag2_detect_frame -- the per-frame
pipeline (grid -> normals -> frames -> sweep -> prune -> images -> LeNet -> select -> top-k) at
fixed maximum shapes, captured in a hipGraph and replayed.

The bar: a replayed frame returns THE SAME BYTES as the step-by-step path
(ag2_set_cloud + ag2_compute_normals + ag2_detect) on the same cloud -- records, scores, order --
and both equal the oracle within the LeNet tolerance (the step-by-step path's own tests)."""
import numpy as np
import pytest

from conftest import scene_params
from agile_grasp2_amd import scene
from agile_grasp2_amd.weights import make_lenet_weights

pytestmark = pytest.mark.gpu


def _clouds(n_frames, n_target, n_samples, kinds=("tabletop",)):
    out = []
    for k in range(n_frames):
        xyz, ws = scene.make_scene(seed=10 + k, n_target=n_target + 137 * (k % 3), kind=kinds[k % len(kinds)])
        idx = scene.draw_samples(20 + k, xyz.shape[0], n_samples)
        out.append((xyz, ws, idx))
    return out


def _pair(ws, **kw):
    from agile_grasp2_amd import capi
    prm = scene_params(ws, **kw)
    w = make_lenet_weights(7)
    d_frame, d_step = capi.Detector(**prm), capi.Detector(**prm)
    for d in (d_frame, d_step):
        d.lenet_load(w)
    return d_frame, d_step


def _stepwise(d, xyz, idx, seed, do_prune=True):
    d.set_cloud(xyz)
    d.compute_normals()
    sel, n_scored = d.detect(sample_idx=idx, seed=seed, do_prune=do_prune, want_all=False)
    return sel, n_scored


@pytest.mark.parametrize("use_graph", [True, False])
def test_frames_equal_the_stepwise_path(use_graph):
    frames = _clouds(7, 20000, 300)
    ws = frames[0][1]   # one workspace for the stream (the scenes share it)
    df, ds = _pair(ws, min_score_diff=-50.0, num_selected=40)
    df.stream_configure(0, 0, use_graph)
    total = 0
    for k, (xyz, _, idx) in enumerate(frames):
        got, gn = df.detect_frame(xyz, idx, seed=k)
        want, wn = _stepwise(ds, xyz, idx, seed=k)
        assert gn == wn and got.tobytes() == want.tobytes(), k
        total += len(want)
    assert total > 50
    fi = df.frame_info()
    assert fi.frames == 7 and fi.stepwise_runs == 1 and fi.fallbacks == 0
    if use_graph:
        assert fi.captures == 1 and fi.graph_replays == 5 and fi.plain_runs == 1 and fi.capture_failed == 0
    else:
        assert fi.captures == 0 and fi.graph_replays == 0 and fi.plain_runs == 6
    # the context is left as after set_cloud + compute_normals + detect of the last frame
    xyz, _, idx = frames[-1]
    assert np.array_equal(df.get_normals().view(np.uint64), ds.get_normals().view(np.uint64))
    gc, sc = df.counters(), ds.counters()
    for f in ("n_frames", "n_hypotheses", "n_scored", "n_selected", "sum_kcrop", "sum_p"):
        assert getattr(gc, f) == getattr(sc, f), f
    df.close()
    ds.close()


def test_frames_of_a_dense_cloud_use_the_renderers_for_large_images():
    """An un-voxelised cloud: most images hold more than 1 024 points, so every frame classifies its images
    by renderer and hands them out through the counters in device memory (k_render_classify, ImgQueue) -- inside
    the captured sequence the counters are cleared by a memset node of the graph.  Replays must return the
    bytes of the step-by-step path, and the step-by-step images those of the oracle."""
    from oracle import api
    xyz, ws = scene.make_scene(seed=4, n_target=120000, kind="objects", voxel=None)
    idxs = [scene.draw_samples(40 + k, xyz.shape[0], 120) for k in range(4)]
    df, ds = _pair(ws, num_orientations=16, min_score_diff=-1e30, num_selected=100000)
    df.stream_configure(0, 0, True)
    big = 0
    for k, idx in enumerate(idxs):
        got, gn = df.detect_frame(xyz, idx, seed=k)
        want, wn = _stepwise(ds, xyz, idx, seed=k)
        assert gn == wn and got.tobytes() == want.tobytes(), k
        big += int((want["n_points"] > 1024).sum())
    assert big > 20          # the large-image renderers ran in every mode
    fi = df.frame_info()
    assert fi.frames == 4 and fi.graph_replays >= 1 and fi.fallbacks == 0
    # the images themselves against the oracle (the step path of the last frame)
    o = api.Oracle(**scene_params(ws, num_orientations=16))
    o.set_cloud(xyz)
    o.compute_normals()
    ho = o.generate_hypotheses(sample_idx=idxs[-1], seed=3)
    hg = ds.generate_hypotheses(sample_idx=idxs[-1], seed=3)
    assert hg.tobytes() == ho.tobytes() and int((ho["n_points"] > 1024).sum()) > 5
    assert np.array_equal(ds.render_images(0, len(ho)), o.render_images(0, len(ho)))
    df.close()
    ds.close()


def test_a_frame_that_needs_the_long_list_stage_after_frames_that_did_not():
    """The captured sequence leaves the sweep's long-list stage out while no frame has needed it.  A frame inside
    the learned shapes whose neighbourhoods are too long for the first stage (a dense blob of points) is noticed
    when its results arrive, repeated step by step, and the sequence is captured again with the stage."""
    frames = _clouds(3, 20000, 200)
    ws = frames[0][1]
    df, ds = _pair(ws, min_score_diff=-1e30, num_selected=40)
    df.stream_configure(0, 0, True)
    for k, (xyz, _, idx) in enumerate(frames):
        got, gn = df.detect_frame(xyz, idx, seed=k)
        want, wn = _stepwise(ds, xyz, idx, seed=k)
        assert gn == wn and got.tobytes() == want.tobytes(), k
    assert df.counters().n_overflow_samples == 0 and df.frame_info().graph_replays >= 1
    rng = np.random.default_rng(5)
    base = frames[0][0]
    centre = base[frames[0][2][0]]
    blob = (centre + rng.uniform(-0.03, 0.03, size=(18000, 3))).astype(np.float32)
    mixed = np.concatenate([base[:2500], blob]).astype(np.float32)
    idx = np.sort(rng.choice(np.arange(2500, mixed.shape[0]), 200, replace=False)).astype(np.int32)
    before = df.frame_info().fallbacks
    for rep in range(3):
        got, gn = df.detect_frame(mixed, idx, seed=9)
        want, wn = _stepwise(ds, mixed, idx, seed=9)
        assert gn == wn and got.tobytes() == want.tobytes(), rep
    assert ds.counters().n_overflow_samples > 0
    fi = df.frame_info()
    assert fi.fallbacks == before + 1 and fi.graph_replays >= 2
    df.close()
    ds.close()


def test_frames_with_grasp_clusters():
    """min_inliers > 0 (the launch files' default is 5): the clustering between the threshold and the top-k is part
    of the captured sequence; replays return the bytes of the step-by-step path."""
    frames = _clouds(5, 20000, 300)
    ws = frames[0][1]
    df, ds = _pair(ws, min_score_diff=-1e30, num_selected=40)
    for d in (df, ds):
        d.set_min_inliers(2)
    df.stream_configure(0, 0, True)
    total = 0
    for k, (xyz, _, idx) in enumerate(frames):
        got, gn = df.detect_frame(xyz, idx, seed=k)
        want, wn = _stepwise(ds, xyz, idx, seed=k)
        assert gn == wn and got.tobytes() == want.tobytes(), k
        total += len(want)
    assert total > 10
    fi = df.frame_info()
    assert fi.graph_replays == 3 and fi.fallbacks == 0 and fi.stepwise_runs == 1
    df.close()
    ds.close()


def test_frames_everything_selected_and_no_prune():
    """num_selected < 0 keeps every record above the threshold; do_prune off scores every hypothesis."""
    frames = _clouds(4, 12000, 200, kinds=("tabletop", "objects"))
    ws = frames[0][1]
    df, ds = _pair(ws, min_score_diff=-1e30, num_selected=-1)
    for k, (xyz, _, idx) in enumerate(frames):
        got, gn = df.detect_frame(xyz, idx, seed=3, do_prune=False)
        want, wn = _stepwise(ds, xyz, idx, seed=3, do_prune=False)
        assert gn == wn and len(got) == gn and got.tobytes() == want.tobytes(), k
    assert df.frame_info().graph_replays == 2
    df.close()
    ds.close()


def test_a_frame_that_outgrows_the_shapes_is_repeated_stepwise():
    small = _clouds(3, 8000, 100)
    big_xyz, _ = scene.make_scene(seed=77, n_target=30000)
    big_idx = scene.draw_samples(5, big_xyz.shape[0], 250)
    ws = small[0][1]
    df, ds = _pair(ws, min_score_diff=-1e30, num_selected=25)
    seq = [small[0], small[1], small[2], (big_xyz, ws, big_idx), small[0], (big_xyz, ws, big_idx), small[1]]
    for k, (xyz, _, idx) in enumerate(seq):
        got, gn = df.detect_frame(xyz, idx, seed=k)
        want, wn = _stepwise(ds, xyz, idx, seed=k)
        assert gn == wn and got.tobytes() == want.tobytes(), k
    fi = df.frame_info()
    assert fi.stepwise_runs == 2          # the first frame and the first big one
    assert fi.captures == 2 and fi.graph_replays >= 2
    assert fi.max_points >= big_xyz.shape[0] and fi.max_samples == 250
    df.close()
    ds.close()


def test_device_resident_frames():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    frames = _clouds(4, 15000, 200)
    ws = frames[0][1]
    df, ds = _pair(ws, min_score_diff=-1e30, num_selected=30)
    nbytes = max(f[0].nbytes for f in frames)
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), nbytes) == 0
    for k, (xyz, _, idx) in enumerate(frames):
        assert hip.hipMemcpy(dptr, xyz.ctypes.data_as(C.c_void_p), xyz.nbytes, 1) == 0
        got, gn = df.detect_frame(sample_idx=idx, seed=k, dptr=dptr.value, n=xyz.shape[0], stride=12)
        want, wn = _stepwise(ds, xyz, idx, seed=k)
        assert gn == wn and got.tobytes() == want.tobytes(), k
    hip.hipFree(dptr)
    df.close()
    ds.close()


# ---- frames of the RAW sensor cloud: ag2_detect_frame_raw (filter + voxel grid + sub-sampling inside) ----
REC_FIELDS = ("sample_slot", "orientation", "half_antipodal", "full_antipodal", "n_points", "axis",
              "approach", "binormal", "surface", "bottom", "top", "width")


def _oracle_raw_frame(o, raw, num_samples, sample_seed, seed, do_prune=True):
    """GraspDetector::preprocessPointCloud + detectGraspPoses on the oracle (grasp_detector.cpp:285-335, :84-282)."""
    m = o.preprocess_cloud(raw, voxel_size=scene.VOXEL)
    cloud, _ = o.get_cloud()
    idx = o.subsample_uniformly(num_samples, seed=sample_seed)
    o.compute_normals()
    sel, allh = o.detect(sample_idx=idx, seed=seed, do_prune=do_prune)
    return m, cloud, idx, sel, allh


def check_raw_frame_against_oracle(got, n_scored, n_vox, d, oracle_out, tag=""):
    """every scored record of a frame (the detector selects everything: num_selected < 0, no threshold)
    against the oracle's: bytes of every field but the score, scores within the LeNet tolerance; the
    processed cloud and the drawn sample indices byte-equal."""
    m, cloud, idx, _, allh = oracle_out
    assert n_vox == m, (tag, n_vox, m)
    assert n_scored == len(allh) == len(got), (tag, n_scored, len(allh), len(got))
    gx, _ = d.get_cloud()
    assert gx.tobytes() == cloud.tobytes(), tag
    assert np.array_equal(d.get_samples(), idx), tag
    if len(allh) == 0:
        return 0.0
    kg = np.lexsort((got["orientation"], got["sample_slot"]))
    kw = np.lexsort((allh["orientation"], allh["sample_slot"]))
    g, w = got[kg], allh[kw]
    for f in REC_FIELDS:
        if f != "full_antipodal":
            assert np.array_equal(g[f], w[f]), (tag, f)
    assert np.all(g["full_antipodal"] == 1), tag   # a selected hand is marked so, grasp_detector.cpp:205
    tol = 2e-4 * np.abs(w["score"]).max() + 2e-3
    assert np.abs(g["score"] - w["score"]).max() <= tol, tag
    assert np.all(np.diff(got["score"]) <= 0), tag     # the frame's own order: score descending
    return tol


@pytest.mark.parametrize("use_graph", [True, False])
def test_raw_frames_equal_the_oracle_and_the_stepwise_path(use_graph):
    from agile_grasp2_amd import capi
    from oracle import api
    raws, ws0 = scene.make_stream(40, 50000, 6, voxel=None)   # one scene, its objects drifting
    # what a depth sensor delivers besides points: NaN where a pixel has no return, now and then an infinity --
    # in every frame at other places, so that the replayed sequence meets them where the captured one had points
    for k, r in enumerate(raws):
        r[3 + k::97, k % 3] = np.nan
        r[11 + 2 * k::389] = np.inf
        r[17 + k::1013, (k + 1) % 3] = -np.inf
    frames = [(r, ws0) for r in raws]
    ws = np.array(ws0, dtype=np.float64)
    ws[1] -= 0.03   # the workspace filter has something to cut (a strip of the table)
    ws[3] -= 0.02
    prm = scene_params(ws, min_score_diff=-1e30, num_selected=-1)
    w = make_lenet_weights(7)
    df, ds = capi.Detector(**prm), capi.Detector(**prm)
    o = api.Oracle(**dict(prm, num_threads=8))
    for x in (df, ds, o):
        x.lenet_load(w)
    df.stream_configure(0, 0, use_graph)
    ns, total = 250, 0
    for k, (raw, _) in enumerate(frames):
        got, n_sc, n_vox = df.detect_frame_raw(raw, num_samples=ns, sample_seed=100 + k, seed=k)
        assert 0 < n_vox < raw.shape[0]
        # the step-by-step calls on a second context: same bytes
        m = ds.preprocess_cloud(raw, voxel_size=scene.VOXEL)
        k_s = ds.subsample_uniformly(ns, seed=100 + k, want_indices=False)
        ds.compute_normals()
        want, wn = ds.detect(n_resident=k_s, seed=k, do_prune=True, want_all=False)
        assert m == n_vox and wn == n_sc and got.tobytes() == want.tobytes(), k
        check_raw_frame_against_oracle(got, n_sc, n_vox, df, _oracle_raw_frame(o, raw, ns, 100 + k, k), tag=k)
        total += n_sc
    assert total > 100
    fi = df.frame_info()
    assert fi.frames == 6 and fi.stepwise_runs == 1 and fi.fallbacks == 0, [(f, getattr(fi, f)) for f, _ in fi._fields_]
    if use_graph:
        assert fi.captures == 1 and fi.graph_replays == 4 and fi.plain_runs == 1 and fi.capture_failed == 0
    else:
        assert fi.graph_replays == 0 and fi.plain_runs == 5
    # the context is left as after preprocess + subsample + normals + detect of the last frame
    assert np.array_equal(df.get_normals().view(np.uint64), ds.get_normals().view(np.uint64))
    for x in (df, ds):
        x.close()


def test_raw_frames_that_leave_the_shapes():
    """a raw frame with more points, a frame whose lattice outgrows the bitmap (an object far above the
    scene), a cloud with fewer voxels than num_samples (every point is a sample, grasp_detector.cpp:322-330),
    a switch between the two entry points: each runs step by step (or is repeated so) with the same bytes."""
    from agile_grasp2_amd import capi
    small = [scene.make_scene(seed=60 + k, n_target=30000, voxel=None, spacing=0.0015) for k in range(3)]
    big, _ = scene.make_scene(seed=70, n_target=90000, voxel=None, spacing=0.0015)
    ws = np.array(small[0][1], dtype=np.float64)
    ws[5] = 5.0
    tall = np.concatenate([small[1][0], small[1][0][:2000] + np.float32([0.0, 0.0, 2.5])]).astype(np.float32)
    tiny = small[2][0][::400].copy()
    prm = scene_params(ws, min_score_diff=-1e30, num_selected=40)
    w = make_lenet_weights(7)
    df, ds = capi.Detector(**prm), capi.Detector(**prm)
    for x in (df, ds):
        x.lenet_load(w)
    ns = 150
    seq = [small[0][0], small[1][0], small[2][0], big, small[0][0], tall, small[1][0], tiny, small[2][0]]
    for k, raw in enumerate(seq):
        got, n_sc, n_vox = df.detect_frame_raw(raw, num_samples=ns, sample_seed=7, seed=k)
        m = ds.preprocess_cloud(raw, voxel_size=scene.VOXEL)
        k_s = ds.subsample_uniformly(ns, seed=7, want_indices=False)
        ds.compute_normals()
        want, wn = ds.detect(n_resident=k_s, seed=k, do_prune=True, want_all=False)
        assert m == n_vox and wn == n_sc and got.tobytes() == want.tobytes(), k
        assert np.array_equal(df.get_samples(), ds.get_samples()), k
    fi = df.frame_info()
    assert fi.fallbacks >= 1 and fi.graph_replays >= 2 and fi.stepwise_runs >= 4, [(f, getattr(fi, f)) for f, _ in fi._fields_]
    # the other entry point on the same context, and back
    xyz, _, idx = _clouds(1, 9000, 100)[0]
    ds2 = capi.Detector(**prm)
    ds2.lenet_load(w)
    for k in range(3):
        got, gn = df.detect_frame(xyz, idx, seed=k)
        want, wn = _stepwise(ds2, xyz, idx, seed=k)
        assert gn == wn and got.tobytes() == want.tobytes(), k
    got, n_sc, n_vox = df.detect_frame_raw(small[0][0], num_samples=ns, sample_seed=7, seed=0)
    m = ds.preprocess_cloud(small[0][0], voxel_size=scene.VOXEL)
    k_s = ds.subsample_uniformly(ns, seed=7, want_indices=False)
    ds.compute_normals()
    want, wn = ds.detect(n_resident=k_s, seed=0, do_prune=True, want_all=False)
    assert m == n_vox and wn == n_sc and got.tobytes() == want.tobytes()
    for x in (df, ds, ds2):
        x.close()


# ---- the asynchronous form: ag2_submit_frame* / ag2_wait_frame behind ag2_pipe -----------------------------
@pytest.mark.parametrize("raw", [True, False])
def test_pipe_returns_the_bytes_of_the_synchronous_calls(raw):
    """One caller thread, two frames in flight (ag2_pipe, depth 2): every frame's results are byte for byte
    those of the synchronous ag2_detect_frame[_raw] on its own context, in submission order -- clouds handed
    over in host memory (page-locked staging + asynchronous DMA) and resident in HBM."""
    import ctypes as C
    from agile_grasp2_amd import capi
    n_frames, ns = 9, 200
    if raw:
        clouds, ws = scene.make_stream(80, 40000, n_frames, voxel=None)
        idxs = [None] * n_frames
    else:
        clouds, ws = scene.make_stream(81, 15000, n_frames)
        idxs = [scene.draw_samples(90 + k, c.shape[0], ns) for k, c in enumerate(clouds)]
    prm = scene_params(ws, min_score_diff=-1e30, num_selected=35)
    w = make_lenet_weights(7)
    sync = capi.Detector(**prm)
    sync.lenet_load(w)
    want = []
    for k, c in enumerate(clouds):
        if raw:
            sel, n_sc, n_vox = sync.detect_frame_raw(c, num_samples=ns, sample_seed=500 + k, seed=k)
        else:
            sel, n_sc = sync.detect_frame(c, idxs[k], seed=k)
            n_vox = c.shape[0]
        want.append((sel.tobytes(), n_sc, n_vox))
    assert sum(x[1] for x in want) > 100
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    for on_device in (False, True):
        pipe = capi.Pipe(depth=2, **prm)
        pipe.lenet_load(w)
        with pytest.raises(RuntimeError, match="pipe empty"):
            pipe.wait()
        dbufs = []
        if on_device:   # every frame its own device buffer: it must stay valid until the frame's wait
            for c in clouds:
                dp = C.c_void_p()
                assert hip.hipMalloc(C.byref(dp), c.nbytes) == 0
                assert hip.hipMemcpy(dp, c.ctypes.data_as(C.c_void_p), c.nbytes, 1) == 0
                dbufs.append(dp)

        def submit(k):
            kw = dict(dptr=dbufs[k].value, n=clouds[k].shape[0], stride=12) if on_device else dict(xyz=clouds[k])
            if raw:
                pipe.submit_raw(num_samples=ns, sample_seed=500 + k, seed=k, **kw)
            else:
                pipe.submit(sample_idx=idxs[k], seed=k, **kw)

        got = []
        submit(0)
        submit(1)
        with pytest.raises(RuntimeError, match="pipe full"):
            submit(2)
        for k in range(2, n_frames):
            got.append(pipe.wait())
            submit(k)
        got.append(pipe.wait())
        got.append(pipe.wait())
        assert len(got) == n_frames
        for k, (sel, n_sc, n_vox) in enumerate(got):
            assert (sel.tobytes(), n_sc, n_vox) == want[k], (on_device, k)
        pipe.close()
        for dp in dbufs:
            hip.hipFree(dp)
    sync.close()


def test_pipe_frames_of_different_sizes_and_a_short_result_buffer():
    """(1) the result capacity belongs to the frame that is waited for, not to the last one
    submitted -- two frames with very different sample counts in flight, everything selected
    (num_selected = -1), the small one submitted last; (2) a wait with too small a buffer reports
    AG2_ERR_CAPACITY and KEEPS the frame: a second wait with room brings the same bytes as the synchronous
    call; the pipe does not advance in between."""
    import ctypes as C
    from agile_grasp2_amd import capi
    clouds, ws = scene.make_stream(83, 15000, 4)
    prm = scene_params(ws, min_score_diff=-1e30, num_selected=-1)
    w = make_lenet_weights(7)
    n_min = min(c.shape[0] for c in clouds)
    big = np.ascontiguousarray(scene.draw_samples(5, n_min, 400), dtype=np.int32)
    small = scene.draw_samples(6, n_min, 3)
    sync = capi.Detector(**prm)
    sync.lenet_load(w)
    want = [sync.detect_frame(clouds[k], (big, small)[k % 2], seed=k) for k in range(4)]
    assert len(want[0][0]) > 3 * 8 and len(want[0][0]) > len(want[1][0])   # more records than the small frame's capacity
    pipe = capi.Pipe(depth=2, **prm)
    pipe.lenet_load(w)
    got = []
    pipe.submit(clouds[0], big, seed=0)
    pipe.submit(clouds[1], small, seed=1)      # (round 3: this overwrote the capacity used for frame 0)
    got.append(pipe.wait())
    pipe.submit(clouds[2], big, seed=2)
    got.append(pipe.wait())
    pipe.submit(clouds[3], small, seed=3)
    got.append(pipe.wait())
    got.append(pipe.wait())
    for k in range(4):
        assert got[k][0].tobytes() == want[k][0].tobytes() and got[k][1] == want[k][1], k
    # a short buffer: the frame stays, the second wait gets it
    pipe.submit(clouds[0], big, seed=0)
    buf = np.zeros(len(want[0][0]), dtype=capi.HYP_DTYPE)
    ns, na, nv = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    rc = pipe.L.ag2_pipe_wait(pipe.h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(2), C.byref(ns), C.byref(na), C.byref(nv))
    assert rc == -3 and ns.value == len(want[0][0])       # AG2_ERR_CAPACITY, and how much room it takes
    assert b"kept" in pipe.L.ag2_pipe_last_error(pipe.h)
    rc = pipe.L.ag2_pipe_wait(pipe.h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(buf)), C.byref(ns), C.byref(na),
                              C.byref(nv))
    assert rc == 0 and ns.value == len(buf) and buf.tobytes() == want[0][0].tobytes() and na.value == want[0][1]
    with pytest.raises(RuntimeError, match="pipe empty"):
        pipe.wait()
    pipe.close()
    # the SYNCHRONOUS call with a short buffer: AG2_ERR_CAPACITY, *n_selected = the room it needs, and the context
    # is free for the next call (a kept frame would block it: "a frame is in flight")
    xyz0 = np.ascontiguousarray(clouds[0], dtype=np.float32)
    ns, na = C.c_size_t(0), C.c_size_t(0)
    rc = sync.L.ag2_detect_frame(sync.h, xyz0.ctypes.data_as(C.c_void_p), C.c_int(0), C.c_size_t(xyz0.shape[0]),
                                 C.c_size_t(12), big.ctypes.data_as(C.c_void_p), C.c_size_t(len(big)), C.c_uint64(0),
                                 C.c_int(1), buf.ctypes.data_as(C.c_void_p), C.c_size_t(2), C.byref(ns), C.byref(na))
    assert rc == -3 and ns.value == len(want[0][0])
    again = sync.detect_frame(clouds[0], big, seed=0)
    assert again[0].tobytes() == want[0][0].tobytes()
    sync.close()


def test_wait_modes_return_the_same_bytes_and_report_their_cost():
    """ag2_set_wait_mode: polling with the default spin, polling that yields at once
    (spin_us = 0) and waiting for the stream give the same bytes for frames and for the plain detect step;
    ag2_get_wait_info reports the host time inside the submit and the wait half of a frame and counts the
    yields; a wait that is served by polling never falls back to the stream."""
    from agile_grasp2_amd import capi
    clouds, ws = scene.make_stream(84, 20000, 5)
    idx = scene.draw_samples(7, min(c.shape[0] for c in clouds), 150)
    prm = scene_params(ws, min_score_diff=-1e30, num_selected=25)
    w = make_lenet_weights(7)
    out = {}
    for mode in ("default", "yield", "stream"):
        d = capi.Detector(**prm)
        d.lenet_load(w)
        if mode == "yield":
            d.set_wait_mode(True, 0)
        elif mode == "stream":
            d.set_wait_mode(False, 0)
        res = []
        for k, c in enumerate(clouds):
            sel, n_sc = d.detect_frame(c, idx, seed=k)
            res.append((sel.tobytes(), n_sc))
        wi = d.wait_info()
        assert wi.last_wait_us > 0 and wi.last_submit_us > 0 and wi.poll == (0 if mode == "stream" else 1)
        assert wi.poll_fallbacks == 0
        if mode == "stream":
            assert wi.poll_yields == 0
        if mode == "yield":
            assert wi.poll_yields > 0 and wi.spin_us == 0
        for k in range(3):   # the step-by-step calls poll the same way (extent read-back, results)
            d.set_cloud(clouds[k])
            d.compute_normals()
            sel, n_sc = d.detect(sample_idx=idx, seed=k, want_all=False)
            res.append((sel.tobytes(), n_sc))
        assert d.wait_info().poll_fallbacks == 0
        out[mode] = res
        with pytest.raises(RuntimeError):
            d.set_wait_mode(True, -1)
        d.close()
    assert out["default"] == out["yield"] == out["stream"]
    assert sum(n for _, n in out["default"]) > 50


def test_raw_frames_that_are_empty_or_fully_filtered():
    """A raw frame with no points, one whose points all lie outside the workspace, and an ordinary one in
    between: no error, nothing selected, and the stream goes on (the reference returns an empty list for
    an empty cloud, grasp_detector.cpp:86-91)."""
    from agile_grasp2_amd import capi
    raws, ws = scene.make_stream(95, 30000, 3, voxel=None)
    prm = scene_params(ws, min_score_diff=-1e30, num_selected=20)
    d = capi.Detector(**prm)
    d.lenet_load(make_lenet_weights(7))
    outside = (raws[0] + np.float32([10.0, 0.0, 0.0])).astype(np.float32)
    seq = [raws[0], np.zeros((0, 3), np.float32), raws[1], outside, raws[2], raws[0]]
    counts = []
    for k, raw in enumerate(seq):
        sel, n_sc, n_vox = d.detect_frame_raw(raw, num_samples=150, sample_seed=k, seed=k)
        counts.append((len(sel), n_sc, n_vox))
    assert counts[1] == (0, 0, 0) and counts[3] == (0, 0, 0), counts
    assert all(c[2] > 1000 and c[1] > 0 for i, c in enumerate(counts) if i not in (1, 3)), counts
    # the pipe takes the same frames
    pipe = capi.Pipe(depth=2, **prm)
    pipe.lenet_load(make_lenet_weights(7))
    got = []
    for k, raw in enumerate(seq):
        pipe.submit_raw(raw, num_samples=150, sample_seed=k, seed=k)
        if k >= 1:
            got.append(pipe.wait())
    got.append(pipe.wait())
    assert [(len(s), a, v) for s, a, v in got] == counts
    pipe.close()
    d.close()
