"""The C++ host mirror (include/agile_grasp2/*.h, agile_grasp2_amd/host): same class / method names as
the reference's GraspDetector / HandSearch / Learning / Classifier / CloudCamera.

not-gpu: the library and a driver program compile and link against the headers; the parameter
readers (key=value text, roslaunch XML with the reference's own parameter names) parse.
gpu: the driver runs CloudCamera -> HandSearch::generateHypotheses -> Learning::createGraspImages ->
Classifier::ClassifyBatch and GraspDetector::detectGraspPoses; results must equal the oracle's
(poses/labels/images bit-exact, logits within the fp32 tolerance) and the GraspMsg wire bytes must be
the f64/f32 fields of the selected hypotheses.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import scene_params
from agile_grasp2_amd import scene
from agile_grasp2_amd.weights import make_lenet_weights, save_ag2w

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "agile_grasp2_amd", "host")
CSRC_DIR = os.path.join(ROOT, "agile_grasp2_amd", "csrc")


def build_driver(tmp):
    subprocess.check_call(["make", "-C", CSRC_DIR, "-s", "-j", "8"])
    subprocess.check_call(["make", "-C", HOST_DIR, "-s"])
    exe = os.path.join(tmp, "test_host_api")
    subprocess.check_call([
        "g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"),
        os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp"), "-o", exe,
        "-L", HOST_DIR, "-lag2host", "-L", CSRC_DIR, "-lag2hip",
        f"-Wl,-rpath,{HOST_DIR}", f"-Wl,-rpath,{CSRC_DIR}"])
    return exe


def test_host_library_and_driver_build(tmp_path):
    exe = build_driver(str(tmp_path))
    assert os.path.exists(exe)
    out = subprocess.check_output(["nm", "-DC", "--defined-only", os.path.join(HOST_DIR, "libag2host.so")], text=True)
    for sym in ("GraspDetector::detectGraspPoses", "GraspDetector::preprocessPointCloud",
                "HandSearch::generateHypotheses", "Learning::createGraspImages", "Classifier::ClassifyBatch",
                "CloudCamera::voxelizeCloud", "CloudCamera::filterWorkspace",
                "GraspHypothesis::convertToGraspMsg"):
        assert sym in out, sym
    # the driver reports usage (exit code 2) without touching the GPU
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def build_node_replay(tmp):
    """tests/cpp/node_replay.cpp: the call sequence of src/nodes/grasp_detection_node.cpp against the
    mirror headers -- it compiles and links only if the mirror offers the names the node uses."""
    subprocess.check_call(["make", "-C", CSRC_DIR, "-s", "-j", "8"])
    subprocess.check_call(["make", "-C", HOST_DIR, "-s"])
    exe = os.path.join(tmp, "node_replay")
    subprocess.check_call([
        "g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
        os.path.join(ROOT, "tests", "cpp", "node_replay.cpp"), "-o", exe,
        "-L", HOST_DIR, "-lag2host", "-L", CSRC_DIR, "-lag2hip",
        f"-Wl,-rpath,{HOST_DIR}", f"-Wl,-rpath,{CSRC_DIR}"])
    return exe


def test_node_call_sequence_compiles_against_the_mirror(tmp_path):
    exe = build_node_replay(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr   # nothing touched the GPU


@pytest.mark.gpu
def test_node_call_sequence_runs(tmp_path, small_scene):
    tmp = str(tmp_path)
    exe = build_node_replay(tmp)
    xyz, ws, idx = small_scene
    w = make_lenet_weights(7)
    wpath, lpath = os.path.join(tmp, "w.ag2w"), os.path.join(tmp, "labels.txt")
    save_ag2w(wpath, w)
    open(lpath, "w").write("0\n1\n")
    xyz.astype("<f4").tofile(os.path.join(tmp, "cloud.f32"))
    text = params_text(ws, wpath, lpath, 5) + "voxelize = false\nnum_samples = 150\n"
    open(os.path.join(tmp, "params.txt"), "w").write(text)
    r = subprocess.run([exe, os.path.join(tmp, "cloud.f32"), os.path.join(tmp, "params.txt")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    m = __import__("re").search(r"topic (\d+) grasps; service all=(\d+) radius=(\d+) indices=(\d+)", r.stdout)
    assert m, r.stdout
    topic, all_, radius, indices = map(int, m.groups())
    assert all_ > 0 and indices > 0 and radius >= 0
    # The node's topic path never calls preprocessPointCloud (grasp_detection_node.cpp:123-143), so the
    # CloudCamera it builds has no sample indices and the hand search iterates over nothing
    # (hand_search.cpp:50,118) -- the mirror reproduces that.
    assert topic == 0


def params_text(ws, wpath, lpath, seed):
    cam = [float(v) for v in scene.CAMERA]
    pose = [1.0, 0.0, 0.0, cam[0], 0.0, 1.0, 0.0, cam[1], 0.0, 0.0, 1.0, cam[2], 0.0, 0.0, 0.0, 1.0]
    return "\n".join([
        "# launch/file_detect_grasps.launch values",
        f"workspace = {list(map(float, ws))}",
        f"camera_pose = {pose}",
        "num_orientations = 8", "nn_radius_taubin = 0.01", "nn_radius_hands = 0.1",
        "finger_width = 0.01", "hand_outer_diameter = 0.09", "hand_depth = 0.06", "hand_height = 0.02",
        "init_bite = 0.01", "filter_half_grasps = false", "gripper_width_range = [0.03, 0.08]",
        "antipodal_mode = 1", f"trained_file = {wpath}", f"label_file = {lpath}", "model_file =",
        "min_score_diff = -1e30", "num_selected = 1000", "plot_mode = 0", f"seed = {seed}", ""])


@pytest.mark.gpu
@pytest.mark.parametrize("weights_format", ["ag2w", "caffemodel"])
def test_cpp_host_matches_oracle(tmp_path, small_scene, weights_format):
    from oracle import api
    tmp = str(tmp_path)
    exe = build_driver(tmp)
    xyz, ws, idx = small_scene
    w = make_lenet_weights(7)
    wpath, lpath = os.path.join(tmp, "w." + weights_format), os.path.join(tmp, "labels.txt")
    if weights_format == "ag2w":
        save_ag2w(wpath, w)
    else:  # Classifier's trained_file as the reference passes it (caffe_classifier.cpp:13-14)
        from test_formats import net_new
        open(wpath, "wb").write(net_new(w))
    open(lpath, "w").write("0\n1\n")  # caffe/labels.txt
    xyz.astype("<f4").tofile(os.path.join(tmp, "cloud.f32"))
    idx.astype("<i4").tofile(os.path.join(tmp, "idx.i32"))
    seed = 5
    open(os.path.join(tmp, "params.txt"), "w").write(params_text(ws, wpath, lpath, seed))
    outp = os.path.join(tmp, "out.bin")
    r = subprocess.run([exe, os.path.join(tmp, "cloud.f32"), os.path.join(tmp, "idx.i32"),
                        os.path.join(tmp, "params.txt"), outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "host api ok" in r.stdout and "label0=0" in r.stdout

    prm = scene_params(ws, min_score_diff=-1e30, num_selected=1000)
    o = api.Oracle(**dict(prm, num_threads=4))
    o.set_cloud(xyz)
    o.compute_normals()
    o.lenet_load(w)
    want = o.generate_hypotheses(sample_idx=idx, seed=seed)
    wimgs = o.render_images(0, len(want))
    wlog = o.lenet_forward(wimgs)
    wsel, wall = o.detect(sample_idx=idx, seed=seed, do_prune=True)

    buf = open(outp, "rb").read()
    off = 0

    def take(fmt):
        nonlocal off
        v = struct.unpack_from(fmt, buf, off)
        off += struct.calcsize(fmt)
        return v

    (nh,) = take("<q")
    assert nh == len(want)
    for k in range(nh):
        rec = np.array(take("<20d"))
        (p,) = take("<q")
        h = want[k]
        ref = np.concatenate([h["axis"], h["approach"], h["binormal"], h["surface"], h["bottom"], h["top"],
                              [h["width"], h["half_antipodal"] + 2 * h["full_antipodal"]]])
        assert np.array_equal(rec, ref), k
        assert p == h["n_points"]
    (ni,) = take("<q")
    assert ni == nh
    imgs = np.frombuffer(buf, dtype=np.uint8, count=ni * 10800, offset=off).reshape(ni, 60, 60, 3)
    off += ni * 10800
    assert np.array_equal(imgs, wimgs)
    (npred,) = take("<q")
    assert npred == nh
    logits = np.frombuffer(buf, dtype="<f4", count=2 * npred, offset=off).reshape(npred, 2)
    off += 8 * npred
    tol = 1e-4 * np.abs(wlog).max() + 1e-3
    assert np.abs(logits - wlog).max() <= tol
    (ns,) = take("<q")
    assert ns == len(wsel) and ns > 0
    wire = np.frombuffer(buf, dtype=np.uint8, count=152 * ns, offset=off)
    off += 152 * ns
    slots = np.frombuffer(buf, dtype="<i4", count=2 * ns, offset=off).reshape(ns, 2)
    key = {(int(h["sample_slot"]), int(h["orientation"])): h for h in wsel}
    assert sorted(map(tuple, slots.tolist())) == sorted(key)
    for k in range(ns):
        h = key[tuple(slots[k])]
        d = np.frombuffer(wire[152 * k: 152 * k + 144].tobytes(), dtype="<f8")
        f = np.frombuffer(wire[152 * k + 144: 152 * (k + 1)].tobytes(), dtype="<f4")
        ref = np.concatenate([h["surface"], h["bottom"], h["top"], h["axis"], h["approach"], h["binormal"]])
        assert np.array_equal(d, ref)                       # GraspMsg.msg: 3 Points + 3 Vector3 (f64)
        assert f[0] == np.float32(h["width"])               # std_msgs/Float32 width
        assert abs(f[1] - np.float32(h["score"])) <= 2 * tol


def two_sided_scene(seed, n_boxes=6):
    """A table and boxes sampled on ALL faces, with the analytic outward normals: a full-antipodal grasp
    (antipodal.cpp:54-81) needs surface normals that point at both fingers, which a single-view cloud whose
    normals are flipped towards the viewpoint (hand_search.cpp:88) never has -- the cloud brings its normals
    (CloudCamera(PointCloudNormal), cloud_camera.cpp:4-32)."""
    rng = np.random.default_rng(seed)
    dens = 1.0 / (0.003 ** 2)
    m = int(0.5 * 0.6 * dens)
    pts = [np.stack([rng.uniform(0.5, 1.0, m), rng.uniform(-0.3, 0.3, m), np.full(m, scene.TABLE_Z)], axis=1)]
    nrm = [np.tile([0.0, 0.0, 1.0], (m, 1))]
    for k in range(n_boxes):
        size = np.array([rng.uniform(0.025, 0.05), rng.uniform(0.06, 0.12), rng.uniform(0.08, 0.15)])
        c = np.array([0.58 + 0.07 * k, rng.uniform(-0.2, 0.2), scene.TABLE_Z + 0.5 * size[2]])
        p, n = scene._box(rng, c, size, rng.uniform(0, np.pi), dens)
        pts.append(p)
        nrm.append(n)
    p = np.concatenate(pts)
    p = p + rng.normal(scale=0.0003, size=p.shape)
    ws = np.array([0.45, 1.05, -0.35, 0.35, scene.TABLE_Z - 0.05, 1.0])
    return p.astype(np.float32), np.concatenate(nrm).astype(np.float32), ws


@pytest.mark.gpu
@pytest.mark.parametrize("mode,min_inliers", [(0, 0), (0, 1), (2, 0), (2, 1)])
def test_cpp_detect_modes_none_and_geometric(tmp_path, mode, min_inliers):
    """GraspDetector::detectGraspPoses with antipodal_mode NONE / GEOMETRIC (grasp_detector.cpp:163-252).
    NONE returns the pruned hypotheses as they are -- BEFORE the clustering and the selection (:170-176),
    whatever min_inliers says; GEOMETRIC keeps the full-antipodal ones (:214-221), clusters them when
    min_inliers > 0 (:228-236) and takes the first num_selected (scores are all equal: list order)."""
    from oracle import api
    tmp = str(tmp_path)
    exe = build_driver(tmp)
    xyz, nrm, ws = two_sided_scene(1)
    idx = scene.draw_samples(3, xyz.shape[0], 400)
    xyz.astype("<f4").tofile(os.path.join(tmp, "cloud.f32"))
    nrm.astype("<f4").tofile(os.path.join(tmp, "normals.f32"))
    idx.astype("<i4").tofile(os.path.join(tmp, "idx.i32"))
    seed, num_selected = 5, 7
    text = params_text(ws, "", "", seed).replace("antipodal_mode = 1", f"antipodal_mode = {mode}")
    text = text.replace("num_selected = 1000", f"num_selected = {num_selected}") + f"min_inliers = {min_inliers}\n"
    open(os.path.join(tmp, "params.txt"), "w").write(text)
    outp = os.path.join(tmp, "out.bin")
    r = subprocess.run([exe, "--modes", os.path.join(tmp, "cloud.f32"), os.path.join(tmp, "idx.i32"),
                        os.path.join(tmp, "params.txt"), outp, os.path.join(tmp, "normals.f32")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert f"modes ok: mode {mode}, min_inliers {min_inliers}" in r.stdout

    o = api.Oracle(**dict(scene_params(ws, min_score_diff=-1e30, num_selected=num_selected), num_threads=4))
    o.set_cloud(xyz, normals=nrm.T.astype(np.float64))
    hyps = o.generate_hypotheses(sample_idx=idx, seed=seed)
    keep = o.prune(len(hyps)).astype(bool)
    want = hyps[keep]
    assert len(want) > num_selected
    if mode == 2:
        want = want[want["full_antipodal"] == 1]
        assert len(want) > num_selected
        if min_inliers > 0:
            want = o.find_clusters(want, min_inliers)
            assert len(want) > num_selected
        want = want[:num_selected]
    rec = np.frombuffer(open(outp, "rb").read(), dtype=np.dtype(
        [("slot", "<i4"), ("orient", "<i4"), ("full", "<i4"), ("half", "<i4"), ("score", "<f8"), ("bottom", "<f8", 3)]),
        offset=8)
    (n,) = struct.unpack_from("<q", open(outp, "rb").read(), 0)
    assert n == len(rec) == len(want) and n > 0
    assert np.array_equal(rec["slot"], want["sample_slot"])       # same hands in the same order
    assert np.array_equal(rec["orient"], want["orientation"])
    assert np.array_equal(rec["full"], want["full_antipodal"])
    assert np.array_equal(rec["half"], want["half_antipodal"])
    assert np.array_equal(rec["bottom"], want["bottom"])          # (moved by the clustering where it ran)
    assert np.array_equal(rec["score"], want["score"])


@pytest.mark.gpu
@pytest.mark.parametrize("tiling", ["replicate", "spatial"])
@pytest.mark.parametrize("min_inliers", [0, 1])
def test_cpp_detect_on_several_devices_equals_one_device(tmp_path, small_scene, min_inliers, tiling):
    """GraspDetector::Params::devices: the C++ host's own path to N GPUs -- one context and one host thread per
    entry, the sample list cut into contiguous ranges, the ranks' scored candidates gathered on the first device
    (ag2_gather_selected: a device-to-device copy here, a peer copy over xGMI between GPUs), clustering and
    top-k there.  devices = [0, 0, 0] on the one GPU of the box against the one-context run: the same hands,
    field for field; scores within the LeNet tolerance (ip1's split-K follows the batch size, so the last bits
    of a score may differ between a third of the list and the whole list).
    tiling = spatial (BASELINE configuration 4 from C++): the sample list ordered along the cloud's longest axis
    -- for one device as for three --, ranges of equal summed neighbour counts, every device holding only its
    interval + halo, binned against the whole cloud's minimum; additionally against the oracle run on the list
    sharding.order_samples_by_x gives (the C++ order, cut and tiles are sharding.py's), and every tile smaller
    than the cloud.  Each run calls detectGraspPoses twice on one detector (packed weights, contexts and peers
    are reused): identical hands."""
    from agile_grasp2_amd import sharding
    from oracle import api
    tmp = str(tmp_path)
    exe = build_driver(tmp)
    xyz, ws, idx = small_scene
    w = make_lenet_weights(7)
    wpath, lpath = os.path.join(tmp, "w.ag2w"), os.path.join(tmp, "labels.txt")
    save_ag2w(wpath, w)
    open(lpath, "w").write("0\n1\n")
    xyz.astype("<f4").tofile(os.path.join(tmp, "cloud.f32"))
    idx.astype("<i4").tofile(os.path.join(tmp, "idx.i32"))
    dt = np.dtype([("slot", "<i4"), ("orient", "<i4"), ("full", "<i4"), ("half", "<i4"), ("score", "<f8"), ("bottom", "<f8", 3)])
    out, tiles = {}, {}
    for tag, extra in (("one", ""), ("three", "devices = [0, 0, 0]\n")):
        open(os.path.join(tmp, f"params_{tag}.txt"), "w").write(
            params_text(ws, wpath, lpath, 5) + f"min_inliers = {min_inliers}\ntiling = {tiling}\n" + extra)
        outp = os.path.join(tmp, f"out_{tag}.bin")
        r = subprocess.run([exe, "--modes", os.path.join(tmp, "cloud.f32"), os.path.join(tmp, "idx.i32"),
                            os.path.join(tmp, f"params_{tag}.txt"), outp], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        out[tag] = np.frombuffer(open(outp, "rb").read(), dtype=dt, offset=8)
        tiles[tag] = [int(v) for line in r.stdout.splitlines() if line.startswith("tile_points") for v in line.split()[1:]]
    a, b = out["one"], out["three"]
    assert len(a) == len(b) and len(a) > (5 if min_inliers == 0 else 1)
    ka, kb = np.lexsort((a["orient"], a["slot"])), np.lexsort((b["orient"], b["slot"]))
    for f in ("slot", "orient", "full", "half", "bottom"):
        assert np.array_equal(a[f][ka], b[f][kb]), f
    tol = 1e-4 * np.abs(a["score"]).max() + 2e-3
    assert np.abs(a["score"][ka] - b["score"][kb]).max() <= tol
    assert np.all(np.diff(b["score"]) <= 0)   # the merge's own order: score descending
    assert len(tiles["three"]) == 3 and tiles["one"] == []
    if tiling == "replicate":
        assert tiles["three"] == [xyz.shape[0]] * 3
        return
    # spatial tiles: smaller than the cloud, and exactly sharding.py's
    axis = sharding.longest_axis(xyz)
    ordered = sharding.order_samples_by_x(xyz, idx, axis)
    bounds = sharding.balanced_bounds(sharding.sample_costs(xyz, ordered, 0.1, axis), 3)
    halo = sharding.tile_halo(0.1, 0.01, 0.01)
    want_tiles = [len(sharding.tile_points(xyz, ordered, r, 3, halo, axis, bounds)[0]) for r in range(3)]
    assert tiles["three"] == want_tiles and max(want_tiles) < xyz.shape[0]
    if min_inliers == 0:   # the hands are those of the oracle on the ordered list (slot = position in it)
        prm = scene_params(ws, min_score_diff=-1e30, num_selected=1000)
        o = api.Oracle(**dict(prm, num_threads=4))
        o.set_cloud(xyz)
        o.compute_normals()
        o.lenet_load(w)
        osel, oall = o.detect(sample_idx=ordered, seed=5, do_prune=True)
        assert len(osel) == len(b)
        ko = np.lexsort((osel["orientation"], osel["sample_slot"]))
        assert np.array_equal(osel["sample_slot"][ko], b["slot"][kb]) and np.array_equal(osel["orientation"][ko], b["orient"][kb])
        assert np.array_equal(osel["bottom"][ko], b["bottom"][kb])
        assert np.abs(osel["score"][ko] - b["score"][kb]).max() <= 2e-4 * np.abs(osel["score"]).max() + 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("min_inliers", [0, 1])
def test_cpp_frames_of_the_raw_cloud_equal_the_two_calls(tmp_path, min_inliers):
    """GraspDetector::detectGraspPosesInFrame (ag2_detect_frame_raw behind the mirror: filter + voxel grid +
    sub-sampling + detect in one captured GPU sequence) three times on one raw cloud -- step by step, at fixed
    shapes, as a graph replay -- against preprocessPointCloud + detectGraspPoses on a second detector: the same
    bytes every time.  With min_inliers > 0 the frames run step by step inside the library (clustering is not
    part of the captured sequence) and must agree as well."""
    tmp = str(tmp_path)
    exe = build_driver(tmp)
    raw, ws = scene.make_scene(seed=12, n_target=50000, voxel=None, spacing=0.0015)
    w = make_lenet_weights(7)
    wpath, lpath = os.path.join(tmp, "w.ag2w"), os.path.join(tmp, "labels.txt")
    save_ag2w(wpath, w)
    open(lpath, "w").write("0\n1\n")
    raw.astype("<f4").tofile(os.path.join(tmp, "raw.f32"))
    text = params_text(ws, wpath, lpath, 5) + f"num_samples = 400\nvoxelize = true\nmin_inliers = {min_inliers}\n"
    text = text.replace("num_selected = 1000", "num_selected = 25")
    open(os.path.join(tmp, "params.txt"), "w").write(text)
    outp = os.path.join(tmp, "out.bin")
    r = subprocess.run([exe, "--frames", os.path.join(tmp, "raw.f32"), os.path.join(tmp, "params.txt"), outp],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "frames ok" in r.stdout
    buf = open(outp, "rb").read()
    dt = np.dtype([("slot", "<i4"), ("orient", "<i4"), ("score", "<f8")])
    runs, off = [], 0
    for _ in range(4):
        (n,) = struct.unpack_from("<q", buf, off)
        off += 8
        runs.append(np.frombuffer(buf, dtype=dt, count=n, offset=off).tobytes())
        off += n * dt.itemsize
        assert n > 0 and n <= 25
    assert runs[0] == runs[1] == runs[2] == runs[3]


def test_launch_xml_and_keyvalue_readers(tmp_path):
    """Params readers accept the reference's own launch file text (parameter NAMES are the
    contract; the file is read at test time from /root/reference when present, else a literal
    excerpt of its parameter names is used)."""
    src = r'''
#include <cstdio>
#include <fstream>
#include <iterator>
#include "agile_grasp2/grasp_detector.h"
int main(int argc, char** argv) {
  std::ifstream f(argv[1]);
  std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  GraspDetector::Params p; std::string err;
  bool ok = (std::string(argv[2]) == "xml") ? GraspDetector::Params::fromLaunchXml(text, &p, &err)
                                            : GraspDetector::Params::fromKeyValueText(text, &p, &err);
  if (!ok) { printf("ERR %s\n", err.c_str()); return 1; }
  printf("%d %d %g %g %d %g %zu %g %d %d %g\n", p.num_samples, p.num_orientations, p.init_bite,
         p.min_score_diff, p.num_selected, p.workspace.size() ? p.workspace[1] : -1.0, p.workspace.size(),
         p.gripper_width_range[1], (int)p.filter_half_grasps, p.min_inliers, p.nn_radius_hands);
  return 0;
}
'''
    tmp = str(tmp_path)
    subprocess.check_call(["make", "-C", CSRC_DIR, "-s", "-j", "8"])
    subprocess.check_call(["make", "-C", HOST_DIR, "-s"])
    cpp = os.path.join(tmp, "p.cpp")
    open(cpp, "w").write(src)
    exe = os.path.join(tmp, "p")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), cpp, "-o", exe,
                           "-L", HOST_DIR, "-lag2host", "-L", CSRC_DIR, "-lag2hip",
                           f"-Wl,-rpath,{HOST_DIR}", f"-Wl,-rpath,{CSRC_DIR}"])
    xml = '''<launch><node name="detect_grasps_file" pkg="agile_grasp2" type="detect_grasps_file">
    <param name="cloud_type" value="0" /> <!-- a comment <param name="bogus" value="1"/> -->
    <rosparam param="workspace"> [0.48, 0.92, -0.6, 0.4, -0.25, 1] </rosparam>
    <rosparam param="camera_pose"> [] </rosparam>
    <param name="num_samples" value="5000" /> <param name="num_threads" value="4" />
    <param name="nn_radius_taubin" value="0.01" /> <param name="nn_radius_hands" value="0.1" />
    <param name="num_orientations" value="8" /> <param name="antipodal_mode" value="1" />
    <param name="voxelize" value="true"/> <param name="filter_half_grasps" value="false"/>
    <param name="gripper_width_range" value="[0.03, 0.08]" />
    <param name="finger_width" value="0.01" /> <param name="hand_outer_diameter" value="0.09" />
    <param name="hand_depth" value="0.06" /> <param name="hand_height" value="0.02" />
    <param name="init_bite" value="0.01" />
    <param name="model_file" value="$(find agile_grasp2)/caffe/test_1batch2.prototxt" />
    <param name="min_score_diff" value="300" /> <param name="batch_size" value="100" />
    <param name="min_inliers" value="5" /> <param name="num_selected" value="30" />
    </node></launch>'''
    ref_launch = "/root/reference/launch/file_detect_grasps.launch"
    if os.path.exists(ref_launch):
        xml = open(ref_launch).read()  # read as data at test time; nothing is copied into the repo
    lx = os.path.join(tmp, "l.launch")
    open(lx, "w").write(xml)
    out = subprocess.check_output([exe, lx, "xml"], text=True).split()
    assert out == ["5000", "8", "0.01", "300", "30", "0.92", "6", "0.08", "0", "5", "0.1"], out
    kv = os.path.join(tmp, "k.txt")
    open(kv, "w").write("num_samples=77 # comment\ninit_bite = 0.02\nworkspace=[1,2,3,4,5,6]\n")
    out = subprocess.check_output([exe, kv, "kv"], text=True).split()
    assert out[0] == "77" and out[2] == "0.02" and out[5] == "2" and out[6] == "6"
    open(kv, "w").write("no_such_parameter = 1\n")
    r = subprocess.run([exe, kv, "kv"], capture_output=True, text=True)
    assert r.returncode == 1 and "unknown parameter" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("min_inliers", [0, 1])
def test_cpp_preprocess_then_detect_matches_oracle(tmp_path, min_inliers):
    """GraspDetector::preprocessPointCloud (GPU front end behind the reference's method name) followed
    by detectGraspPoses on a raw cloud == oracle preprocess -> subsample -> detect [-> findClusters when
    the min_inliers parameter is set, grasp_detector.cpp:228-236]."""
    from oracle import api
    tmp = str(tmp_path)
    exe = build_driver(tmp)
    raw, ws = scene.make_scene(seed=12, n_target=50000, voxel=None, spacing=0.0015)
    w = make_lenet_weights(7)
    wpath, lpath = os.path.join(tmp, "w.ag2w"), os.path.join(tmp, "labels.txt")
    save_ag2w(wpath, w)
    open(lpath, "w").write("0\n1\n")
    raw.astype("<f4").tofile(os.path.join(tmp, "raw.f32"))
    seed, ns_req = 5, 400
    text = params_text(ws, wpath, lpath, seed) + (f"num_samples = {ns_req}\nvoxelize = true\n"
                                                  f"min_inliers = {min_inliers}\n")
    open(os.path.join(tmp, "params.txt"), "w").write(text)
    outp = os.path.join(tmp, "out.bin")
    r = subprocess.run([exe, "--preprocess", os.path.join(tmp, "raw.f32"), os.path.join(tmp, "params.txt"), outp],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "preprocess ok" in r.stdout

    o = api.Oracle(**scene_params(ws, min_score_diff=-1e30, num_selected=1000, num_threads=4))
    m = o.preprocess_cloud(raw, voxel_size=0.003)
    want_xyz, _ = o.get_cloud()
    idx = o.subsample_uniformly(ns_req, seed=seed)
    o.compute_normals()
    o.lenet_load(w)
    o.set_min_inliers(min_inliers)
    wsel, _ = o.detect(sample_idx=idx, seed=seed, do_prune=True)

    buf = open(outp, "rb").read()
    (gm,) = struct.unpack_from("<q", buf, 0)
    off = 8
    assert gm == m
    xyz = np.frombuffer(buf, dtype="<f4", count=3 * gm, offset=off).reshape(gm, 3)
    off += 12 * gm
    assert xyz.tobytes() == want_xyz.tobytes()
    (gk,) = struct.unpack_from("<q", buf, off)
    off += 8
    got_idx = np.frombuffer(buf, dtype="<i4", count=gk, offset=off)
    off += 4 * gk
    assert np.array_equal(got_idx, idx)
    (gs,) = struct.unpack_from("<q", buf, off)
    off += 8
    assert gs == len(wsel) and gs > 0
    rec = np.frombuffer(buf, dtype=np.dtype([("slot", "<i4"), ("orient", "<i4"), ("score", "<f8")]), count=gs, offset=off)
    key = {(int(h["sample_slot"]), int(h["orientation"])): float(h["score"]) for h in wsel}
    assert sorted(zip(rec["slot"].tolist(), rec["orient"].tolist())) == sorted(key)
    tol = 1e-4 * max(abs(v) for v in key.values()) + 2e-3
    for s, q, sc in zip(rec["slot"].tolist(), rec["orient"].tolist(), rec["score"].tolist()):
        assert abs(sc - key[(s, q)]) <= tol


@pytest.mark.gpu
def test_cpp_importance_sampling_rounds_match_oracle(tmp_path, small_scene):
    """ImportanceSampling::detectGraspPoses (importance_sampling.cpp:30-118): the initial round on the
    sample indices, then rounds of xyz samples drawn around the grasps found so far, every round on the
    cloud / grid / normals already resident on the GPU.  The driver dumps each round's samples; the
    oracle, fed the same samples, must return the same hands (the xyz-sample frame path,
    hand_search.cpp:238-317)."""
    from oracle import api
    tmp = str(tmp_path)
    exe = build_driver(tmp)
    xyz, ws, idx = small_scene
    w = make_lenet_weights(7)
    wpath, lpath = os.path.join(tmp, "w.ag2w"), os.path.join(tmp, "labels.txt")
    save_ag2w(wpath, w)
    open(lpath, "w").write("0\n1\n")
    xyz.astype("<f4").tofile(os.path.join(tmp, "cloud.f32"))
    idx.astype("<i4").tofile(os.path.join(tmp, "idx.i32"))
    seed = 9
    open(os.path.join(tmp, "params.txt"), "w").write(params_text(ws, wpath, lpath, seed))
    outp = os.path.join(tmp, "out.bin")
    r = subprocess.run([exe, "--importance", os.path.join(tmp, "cloud.f32"), os.path.join(tmp, "idx.i32"),
                        os.path.join(tmp, "params.txt"), outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "importance ok" in r.stdout

    buf = open(outp, "rb").read()
    n0, nr = struct.unpack_from("<qq", buf, 0)
    off = 16
    rounds = []
    for _ in range(nr):
        (s,) = struct.unpack_from("<q", buf, off)
        off += 8
        rounds.append(np.frombuffer(buf, dtype="<f8", count=3 * s, offset=off).reshape(s, 3).T.copy())
        off += 24 * s
    (nh,) = struct.unpack_from("<q", buf, off)
    off += 8
    rec = np.frombuffer(buf, dtype=np.dtype([("slot", "<i4"), ("orient", "<i4"), ("score", "<f8"),
                                             ("bottom", "<f8", 3)]), count=nh, offset=off)
    assert nr == 3 and all(m.shape == (3, 40) for m in rounds) and n0 > 0

    o = api.Oracle(**scene_params(ws, min_score_diff=-1e30, num_selected=1000, num_threads=4))
    o.set_cloud(xyz)
    o.compute_normals()
    o.lenet_load(w)
    want = [o.detect(sample_idx=idx, seed=seed)[0]] + [o.detect(sample_xyz=m, seed=seed)[0] for m in rounds]
    assert len(want[0]) == n0
    assert sum(len(x) for x in want) == nh and sum(len(x) for x in want[1:]) > 0
    tol = 1e-4 * max(np.abs(x["score"]).max() for x in want if len(x)) + 2e-3
    pos = 0
    for x in want:  # per round: same set of hands, bit-equal positions, scores within the fp32 tolerance
        got = rec[pos: pos + len(x)]
        pos += len(x)
        kg = np.lexsort((got["orient"], got["slot"]))
        kw = np.lexsort((x["orientation"], x["sample_slot"]))
        assert np.array_equal(got["slot"][kg], x["sample_slot"][kw])
        assert np.array_equal(got["orient"][kg], x["orientation"][kw])
        assert np.array_equal(got["bottom"][kg], x["bottom"][kw])
        assert np.abs(got["score"][kg] - x["score"][kw]).max() <= tol
    # samples: 70 % around known grasp surfaces (sigma = 0.02), 30 % cloud points (importance_sampling.cpp:50-90)
    for m in rounds:
        d = np.linalg.norm(m[:, 28:, None] - xyz.T[:, None, :].astype(np.float64), axis=0).min(axis=1)
        assert np.all(d == 0.0)
        assert np.all(np.isfinite(m))
        lo, hi = xyz.min(axis=0) - 0.2, xyz.max(axis=0) + 0.2
        assert np.all(m[:, :28].T > lo) and np.all(m[:, :28].T < hi)
