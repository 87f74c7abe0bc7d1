"""Seeded LeNet weights for the agile_grasp2 classifier architecture.

The trained network of the reference is not in its tree (.MISSING_LARGE_BLOBS lists
caffe/bottles_boxes_cans_5xNeg.caffemodel), so scores cannot be reproduced; parity of the forward
pass is pinned with seeded weights of the same shapes (caffe/test_1batch2.prototxt:1-92), drawn
with the prototxt's own "xavier" filler rule (uniform +-sqrt(3 / fan_in)), biases small uniform.
Blob order is Caffe's: convolution OIHW, inner product out x in.
"""
from __future__ import annotations

import numpy as np

SHAPES = {
    "conv1_w": (20, 3, 5, 5), "conv1_b": (20,),
    "conv2_w": (50, 20, 5, 5), "conv2_b": (50,),
    "ip1_w": (500, 7200), "ip1_b": (500,),
    "ip2_w": (2, 500), "ip2_b": (2,),
}
ORDER = ["conv1_w", "conv1_b", "conv2_w", "conv2_b", "ip1_w", "ip1_b", "ip2_w", "ip2_b"]


def save_ag2w(path: str, w: dict) -> None:
    """Write the flat weight container the C++ Classifier reads (caffe_classifier.h): magic "AG2W"
    followed by the eight blobs as little-endian float32, Caffe blob order."""
    with open(path, "wb") as f:
        f.write(b"AG2W")
        for name in ORDER:
            a = np.ascontiguousarray(w[name], dtype="<f4")
            assert a.shape == SHAPES[name], name
            f.write(a.tobytes())


def make_lenet_weights(seed: int) -> dict:
    rng = np.random.default_rng(seed)
    w = {}
    for name in ORDER:
        shape = SHAPES[name]
        if name.endswith("_w"):
            fan_in = int(np.prod(shape[1:]))
            a = np.sqrt(3.0 / fan_in)
            w[name] = rng.uniform(-a, a, size=shape).astype(np.float32)
        else:
            w[name] = rng.uniform(-0.1, 0.1, size=shape).astype(np.float32)
    return w
