"""The known answers the reference itself holds for the hot path -- it has no
tests, fixtures or golden vectors, only structural constants -- asserted against the oracle, the
C-ABI library and the C++ host mirror (no GPU needed: everything here is host-side derivation).

  * finger slots for the launch-file hand (od 0.09, fw 0.01): FingerHand's constructor,
    src/agile_grasp2/finger_hand.cpp:7-12 -> {-0.08 + 0.08 k/9} u {0.08 k/9}, k = 0..9
  * hand orientations: hand_search.cpp:179-180 (LinSpaced(R+1, -pi/2, pi/2), first R)
  * deepenHand's depth sequence: finger_hand.cpp:118-122 (f64 accumulation of 0.005 steps)
  * the 2-camera Baxter poses: grasp_detector.cpp:113-125 (base_tf, sqrt_tf literals)
  * LeNet layer shapes: caffe/test_1batch2.prototxt:1-92
  * HandSearch's 7-argument constructor: include/agile_grasp2/hand_search.h:114-118

The literals below are data copied out of those lines (a fixture).  When /root/reference is present
(this container; never the GPU box) the test also re-reads the reference's files as text and checks
that the fixture still says what they say.
"""
import os
import re
import subprocess

import numpy as np
import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BASE_TF = np.array([[0, 0.445417, 0.895323, 0.215], [1, 0, 0, -0.015], [0, 0.895323, -0.445417, 0.23], [0, 0, 0, 1.0]])
SQRT_TF = np.array([[0.9366, -0.0162, 0.3500, -0.2863], [0.0151, 0.9999, 0.0058, 0.0058],
                    [-0.3501, -0.0002, 0.9367, 0.0554], [0, 0, 0, 1.0]])
# (name, type, num_output / kernel / stride) in file order
LENET_LAYERS = [("conv1", "Convolution", dict(num_output=20, kernel_size=5)),
                ("pool1", "Pooling", dict(pool="MAX", kernel_size=2, stride=2)),
                ("conv2", "Convolution", dict(num_output=50, kernel_size=5)),
                ("pool2", "Pooling", dict(pool="MAX", kernel_size=2, stride=2)),
                ("ip1", "InnerProduct", dict(num_output=500)),
                ("relu1", "ReLU", {}),
                ("ip2", "InnerProduct", dict(num_output=2)),
                ("prob", "Softmax", {})]
LENET_INPUT = (1, 3, 60, 60)
N_SLOTS_PER_SIDE = 10
LAUNCH_HAND = dict(finger_width=0.01, hand_outer_diameter=0.09, hand_depth=0.06, hand_height=0.02,
                   init_bite=0.01, num_orientations=8)   # launch/file_detect_grasps.launch:20-34


def expected_finger_spacing(od, fw, n=N_SLOTS_PER_SIDE):
    """Eigen's LinSpaced(n, 0, od - fw)(k) = 0 + k * ((od - fw) / (n - 1)), then the two halves."""
    half = np.array([k * ((od - fw) / (n - 1)) for k in range(n)])
    return np.concatenate([half - od + fw, half])


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only in the build container")
def test_fixture_literals_are_what_the_reference_files_say():
    txt = open(os.path.join(REF, "src/agile_grasp2/grasp_detector.cpp")).read()
    for name, want in (("base_tf", BASE_TF), ("sqrt_tf", SQRT_TF)):
        m = re.search(name + r"\s*<<([^;]*);", txt)
        got = np.array([float(v) for v in m.group(1).replace("\n", " ").split(",")]).reshape(4, 4)
        assert np.array_equal(got, want), name
    fh = open(os.path.join(REF, "src/agile_grasp2/finger_hand.cpp")).read()
    assert re.search(r"int n = 10;", fh) and "setLinSpaced(n, 0.0, hand_outer_diameter - finger_width)" in fh
    assert "(fs_half.array() - hand_outer_diameter_ + finger_width_).matrix(), fs_half" in fh
    proto = open(os.path.join(REF, "caffe/test_1batch2.prototxt")).read()
    assert tuple(int(v) for v in re.findall(r"dim:\s*(\d+)", proto)) == LENET_INPUT
    layers = re.findall(r'layer\s*\{\s*name:\s*"(\w+)"\s*type:\s*"(\w+)"(.*?)(?=layer\s*\{|\Z)', proto, flags=re.S)
    assert [(n, t) for n, t, _ in layers] == [(n, t) for n, t, _ in LENET_LAYERS]
    for (n, t, body), (_, _, want) in zip(layers, LENET_LAYERS):
        for key, val in want.items():
            m = re.search(key + r":\s*(\w+)", body)
            assert m and m.group(1) == str(val), (n, key)
    hs = open(os.path.join(REF, "include/agile_grasp2/hand_search.h")).read()
    assert "nn_radius_taubin_(0.03)" in hs and "nn_radius_hands_(0.08)" in hs
    launch = open(os.path.join(REF, "launch/file_detect_grasps.launch")).read()
    for key, val in LAUNCH_HAND.items():
        m = re.search(r'name="%s"\s+value="([^"]+)"' % key, launch)
        assert m and float(m.group(1)) == float(val), key


def test_finger_slots_angles_depths_oracle_and_library():
    from agile_grasp2_amd import capi
    from oracle import api
    fs_o, ang_o, dep_o = api.hand_constants(**LAUNCH_HAND)
    fs_g, ang_g, dep_g = capi.hand_constants(**LAUNCH_HAND)
    # {-0.08 + 0.08 k/9} u {0.08 k/9}
    k = np.arange(10)
    assert np.allclose(fs_o, np.concatenate([-0.08 + 0.08 * k / 9, 0.08 * k / 9]), rtol=0, atol=1e-15)
    want = expected_finger_spacing(0.09, 0.01)
    assert np.array_equal(fs_o, want) and np.array_equal(fs_g, want)      # bit-exact, both sides
    # R = 8: -pi/2 + i * pi/8
    assert np.allclose(ang_o, -np.pi / 2 + np.arange(8) * np.pi / 8, rtol=0, atol=1e-15)
    assert np.array_equal(ang_o, ang_g)
    # for (d = 0.01 + 0.005; d <= 0.06; d += 0.005): 0.015 ... accumulated in f64
    d, seq = 0.01 + 0.005, []
    while d <= 0.06:
        seq.append(d)
        d += 0.005
    assert np.array_equal(dep_o, np.array(seq)) and np.array_equal(dep_g, dep_o)
    assert len(seq) in (9, 10)   # whether 0.06 itself is reached depends on the f64 accumulation: replicated, not assumed
    # other geometries: both sides derive the same bits
    for kw in (dict(finger_width=0.012, hand_outer_diameter=0.105, hand_depth=0.07, init_bite=0.015, num_orientations=16),
               dict(finger_width=0.005, hand_outer_diameter=0.12, hand_depth=0.05, init_bite=0.02, num_orientations=5)):
        a, b = api.hand_constants(**kw), capi.hand_constants(**kw)
        assert np.array_equal(a[0], expected_finger_spacing(kw["hand_outer_diameter"], kw["finger_width"]))
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_mirror_default_camera_poses_and_seven_argument_constructor(tmp_path):
    from test_cpp_host import build_driver
    exe = build_driver(str(tmp_path))
    out = tmp_path / "constants.txt"
    subprocess.check_call([exe, "--constants", str(out)])   # no GPU touched
    rows = {ln.split()[0]: np.array([float(v) for v in ln.split()[1:]]) for ln in open(out) if ln.strip()}
    # grasp_detector.cpp:123-124: left = base * sqrt^-1, right = base * sqrt
    assert np.allclose(rows["cam_tf_right"].reshape(4, 4), BASE_TF @ SQRT_TF, rtol=0, atol=1e-15)
    assert np.allclose(rows["cam_tf_left"].reshape(4, 4), BASE_TF @ np.linalg.inv(SQRT_TF), rtol=0, atol=1e-12)
    assert np.array_equal(rows["finger_spacing"], expected_finger_spacing(0.09, 0.01))
    assert len(rows["angles"]) == 8 and len(rows["depths"]) >= 9
    assert rows["seven_arg_ctor"].tolist() == [0.03, 0.08, 4.0, 500.0]   # hand_search.h:114-118


def test_lenet_layer_shapes():
    """Blob shapes that follow from caffe/test_1batch2.prototxt: what the weight generator, the
    oracle and ag2_lenet_load (Caffe blob order: OIHW, fc as out x in) all assume."""
    from agile_grasp2_amd.weights import make_lenet_weights
    from oracle import api
    c, h, w = LENET_INPUT[1:]
    shapes, flops = {}, 0
    for name, typ, prm in LENET_LAYERS:
        if typ == "Convolution":
            k, o = prm["kernel_size"], prm["num_output"]
            shapes[name + "_w"], shapes[name + "_b"] = (o, c, k, k), (o,)
            h, w = h - k + 1, w - k + 1
            flops += 2 * o * h * w * c * k * k
            c = o
        elif typ == "Pooling":
            h, w = -(-(h - prm["kernel_size"]) // prm["stride"]) + 1, -(-(w - prm["kernel_size"]) // prm["stride"]) + 1
        elif typ == "InnerProduct":
            o = prm["num_output"]
            shapes[name + "_w"], shapes[name + "_b"] = (o, c * h * w), (o,)
            flops += 2 * o * c * h * w
            c, h, w = o, 1, 1
    assert shapes["ip1_w"] == (500, 7200) and shapes["conv2_w"] == (50, 20, 5, 5)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 3628072
    assert abs(flops - 45.41e6) < 0.01e6
    wts = make_lenet_weights(7)
    assert {k: tuple(v.shape) for k, v in wts.items()} == shapes
    o = api.Oracle()
    o.lenet_load(wts)
    out = o.lenet_forward(np.zeros((2,) + (LENET_INPUT[2], LENET_INPUT[3], LENET_INPUT[1]), np.uint8))
    assert out.shape == (2, 2)   # blob ip2 (caffe_classifier.cpp:121), two classes (caffe/labels.txt)
    # zero image: ip2 = W2 relu(b1 ...) -- conv(0) = bias, so the logits are finite and weight-dependent
    assert np.isfinite(out).all() and np.array_equal(out[0], out[1])
