"""On-disk formats either side of the path: the .caffemodel weight reader
(Classifier ctor, caffe_classifier.cpp:13-14) and the PCD reader (CloudCamera(filename),
cloud_camera.cpp:231-240), both in the C++ host mirror.  The reference ships neither a caffemodel nor
a .pcd, so the files are written here: a NetParameter in protobuf wire format with field numbers from
BVLC caffe.proto (parity unpinned against a real model file), and PCD v0.7 ASCII / binary files."""
import os
import struct
import subprocess

import numpy as np
import pytest

from agile_grasp2_amd.weights import make_lenet_weights, save_ag2w
from test_cpp_host import build_driver

NAMES = ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "ip1_w", "ip1_b", "ip2_w", "ip2_b")


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def key(field, wire):
    return varint((field << 3) | wire)


def ld(field, payload):
    return key(field, 2) + varint(len(payload)) + payload


def blob_new(a, packed=True):
    shape = ld(1, b"".join(varint(d) for d in a.shape))                    # BlobShape.dim, packed
    flat = np.ascontiguousarray(a, dtype="<f4").ravel()
    data = ld(5, flat.tobytes()) if packed else b"".join(key(5, 5) + struct.pack("<f", v) for v in flat)
    return ld(7, shape) + data                                              # BlobProto.shape, .data


def blob_v1(a, as_double=False):
    dims = (list(a.shape) + [1, 1, 1, 1])[:4] if a.ndim > 1 else [1, 1, 1, a.shape[0]]
    head = b"".join(key(f, 0) + varint(d) for f, d in zip((1, 2, 3, 4), dims))  # num channels height width
    flat = np.ascontiguousarray(a).ravel()
    if as_double:
        return head + ld(8, flat.astype("<f8").tobytes())                  # double_data, packed
    return head + ld(5, flat.astype("<f4").tobytes())


def net_new(w, unpacked_small=False):
    """NetParameter with `layer` (field 100) records, as current Caffe writes them."""
    msg = ld(1, b"LeNet") + key(3, 0) + varint(0)                           # name, force_backward
    msg += ld(101, key(1, 0) + varint(1))                                   # NetState (skipped)
    for name, typ in (("conv1", "Convolution"), ("pool1", "Pooling"), ("conv2", "Convolution"),
                      ("pool2", "Pooling"), ("ip1", "InnerProduct"), ("relu1", "ReLU"),
                      ("ip2", "InnerProduct"), ("prob", "Softmax")):
        layer = ld(1, name.encode()) + ld(2, typ.encode()) + ld(3, b"data") + ld(4, name.encode())
        if name + "_w" in w:
            for suffix in ("_w", "_b"):
                a = w[name + suffix]
                layer += ld(7, blob_new(a, packed=not (unpacked_small and a.size <= 1520)))
            layer += ld(106, key(1, 0) + varint(20))                        # convolution_param (skipped)
        msg += ld(100, layer)
    return msg


def net_v1(w):
    """Legacy V1LayerParameter records (field 2): name = 4, type enum = 5, blobs = 6."""
    msg = ld(1, b"LeNet")
    for name, enum in (("conv1", 4), ("pool1", 17), ("conv2", 4), ("ip1", 14), ("relu1", 18), ("ip2", 14)):
        layer = ld(4, name.encode()) + key(5, 0) + varint(enum)
        if name + "_w" in w:
            layer += ld(6, blob_v1(w[name + "_w"])) + ld(6, blob_v1(w[name + "_b"], as_double=True))
        msg += ld(2, layer)
    return msg


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    return build_driver(str(tmp_path_factory.mktemp("fmt")))


def read_blobs(driver, path, tmp):
    outp = os.path.join(tmp, "blobs.bin")
    r = subprocess.run([driver, "--caffemodel", path, outp], capture_output=True, text=True, timeout=120)
    return r, (np.fromfile(outp, dtype="<f4") if r.returncode == 0 else None)


@pytest.mark.parametrize("style", ["layer_packed", "layer_unpacked_small", "v1_layers"])
def test_caffemodel_reader_round_trip(driver, tmp_path, style):
    w = make_lenet_weights(11)
    want = np.concatenate([np.asarray(w[k], dtype=np.float32).ravel() for k in NAMES])
    raw = {"layer_packed": lambda: net_new(w), "layer_unpacked_small": lambda: net_new(w, True),
           "v1_layers": lambda: net_v1(w)}[style]()
    path = os.path.join(str(tmp_path), "lenet.caffemodel")
    open(path, "wb").write(raw)
    r, got = read_blobs(driver, path, str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert got.shape == want.shape == (3628072,)
    assert np.array_equal(got, want)   # double_data biases are float-representable: exact as well


def test_caffemodel_reader_rejects_bad_files(driver, tmp_path):
    tmp = str(tmp_path)
    w = make_lenet_weights(11)
    good = net_new(w)
    cases = {
        "truncated": good[: len(good) // 2],
        "garbage": bytes(np.random.default_rng(0).integers(0, 256, 4096, dtype=np.uint8)),
        "wrong_arch": net_new(dict(w, conv1_w=np.zeros((16, 3, 5, 5), np.float32))),
        "missing_layer": net_new({k: v for k, v in w.items() if not k.startswith("ip2")}),
    }
    for name, raw in cases.items():
        path = os.path.join(tmp, name + ".caffemodel")
        open(path, "wb").write(raw)
        r, _ = read_blobs(driver, path, tmp)
        assert r.returncode == 3 and "caffemodel:" in r.stderr, name
    r, _ = read_blobs(driver, os.path.join(tmp, "nope.caffemodel"), tmp)
    assert r.returncode == 3


def write_pcd(path, xyz, rgb, mode, with_normals=False):
    n = xyz.shape[0]
    fields = "x y z rgb" + (" normal_x normal_y normal_z" if with_normals else "")
    k = 7 if with_normals else 4
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n"
            f"FIELDS {fields}\nSIZE {' '.join(['4'] * k)}\nTYPE {' '.join(['F'] * k)}\n"
            f"COUNT {' '.join(['1'] * k)}\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
            f"POINTS {n}\nDATA {mode}\n")
    rec = np.zeros((n, k), dtype="<f4")
    rec[:, :3] = xyz
    rec[:, 3] = rgb
    with open(path, "wb") as f:
        f.write(head.encode())
        if mode == "binary":
            f.write(rec.tobytes())
        else:
            for row in rec:
                f.write((" ".join(repr(float(v)) for v in row) + "\r\n").encode())


@pytest.mark.parametrize("mode,with_normals", [("ascii", False), ("binary", False), ("binary", True)])
def test_pcd_reader(driver, tmp_path, mode, with_normals):
    rng = np.random.default_rng(3)
    xyz = rng.uniform(-1, 1, size=(500, 3)).astype(np.float32)
    xyz[7] = np.nan
    rgb = rng.integers(0, 2 ** 24, size=500).astype(np.uint32).view(np.float32)
    path = os.path.join(str(tmp_path), "c.pcd")
    write_pcd(path, xyz, rgb, mode, with_normals)
    outp = os.path.join(str(tmp_path), "xyz.bin")
    r = subprocess.run([driver, "--pcd", path, outp], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    buf = open(outp, "rb").read()
    (n,) = struct.unpack_from("<q", buf, 0)
    assert n == 500
    got = np.frombuffer(buf, dtype="<f4", count=1500, offset=8).reshape(500, 3)
    assert np.array_equal(got, xyz, equal_nan=True)       # repr(float) round-trips float32 exactly
    rows, cols = struct.unpack_from("<qq", buf, 8 + 6000)
    assert (rows, cols) == (1, 500)                        # cloud_camera.cpp:59: one camera, all ones


def test_ag2w_container_still_read(driver, tmp_path):
    w = make_lenet_weights(5)
    path = os.path.join(str(tmp_path), "w.ag2w")
    save_ag2w(path, w)
    r, _ = read_blobs(driver, path, str(tmp_path))
    assert r.returncode == 3            # readCaffeModel itself refuses the flat container ...
    # ... the Classifier constructor dispatches on the magic (covered by tests/test_cpp_host.py)
