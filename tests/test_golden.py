"""Committed golden vectors (tests/golden/golden_small.npz, made by tests/golden/make_golden.py).

not-gpu: the oracle still reproduces them (guards the oracle against silent drift).
gpu:     the HIP path reproduces them through the C-ABI, with no oracle in the loop.
Bars: integer / index / byte outputs and all f64 pose fields bit-exact; LeNet logits and scores
within 1e-4 * max|logit| + 1e-3 (fp32, different summation order).
"""
import hashlib
import json
import os

import numpy as np
import pytest

from agile_grasp2_amd.weights import make_lenet_weights

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_small.npz")
EXACT = ("sample_slot", "orientation", "half_antipodal", "full_antipodal", "n_points",
         "axis", "approach", "binormal", "surface", "bottom", "top", "width")


@pytest.fixture(scope="module")
def gold():
    g = np.load(GOLD, allow_pickle=False)
    d = {k: g[k] for k in g.files}
    d["params"] = json.loads(str(d["params"]))
    return d


def run_backend(x, gold):
    """Runs the whole path on a backend object (oracle or HIP detector); both expose one API."""
    x.set_cloud(gold["xyz"])
    x.compute_normals()
    seed, base, idx = int(gold["seed"]), int(gold["slot_base"]), gold["sample_idx"]
    out = {}
    out["normals_bits"] = x.get_normals().astype(np.float32).view(np.uint32)
    out["frames"], out["frames_valid"] = x.local_frames(sample_idx=idx, slot_base=base, seed=seed)
    out["hyps"] = x.generate_hypotheses(sample_idx=idx, slot_base=base, seed=seed)
    n = len(out["hyps"])
    out["prune_keep"] = x.prune(n)
    imgs = x.render_images(0, n)
    out["images"] = imgs
    out["lists"] = [x.hyp_points(k, int(out["hyps"][k]["n_points"])) for k in range(3)]
    x.lenet_load(make_lenet_weights(int(gold["weight_seed"])))
    out["logits"] = x.lenet_forward(imgs)
    out["selected"], out["scored"] = x.detect(sample_idx=idx, slot_base=base, seed=seed, do_prune=True)
    return out


def check(out, gold):
    assert np.array_equal(out["normals_bits"], gold["normals_bits"])
    assert np.array_equal(out["frames_valid"], gold["frames_valid"])
    assert np.array_equal(out["frames"].view(np.uint64), gold["frames"].view(np.uint64))
    assert len(out["hyps"]) == len(gold["hyps"])
    for f in EXACT:
        assert np.array_equal(out["hyps"][f], gold["hyps"][f]), f
    assert np.array_equal(out["prune_keep"], gold["prune_keep"])
    sha = np.array([hashlib.sha256(im.tobytes()).hexdigest() for im in out["images"]])
    assert np.array_equal(sha, gold["image_sha256"])
    assert np.array_equal(out["images"][:6], gold["images_first"])
    for k in range(3):
        assert np.array_equal(out["lists"][k][0], gold[f"list{k}_pts"])
        assert np.array_equal(out["lists"][k][1], gold[f"list{k}_nrm"], equal_nan=True)
    tol = 1e-4 * np.abs(gold["logits"]).max() + 1e-3
    assert np.abs(out["logits"] - gold["logits"]).max() <= tol
    for name in ("scored", "selected"):
        a, b = out[name], gold[name]
        assert len(a) == len(b)
        if name == "scored":
            for f in EXACT:
                assert np.array_equal(a[f], b[f]), (name, f)
            assert np.abs(a["score"] - b["score"]).max() <= 2 * tol
        else:
            assert sorted(zip(a["sample_slot"], a["orientation"])) == sorted(zip(b["sample_slot"], b["orientation"]))


def test_oracle_reproduces_golden(gold):
    from oracle import api
    check(run_backend(api.Oracle(**gold["params"]), gold), gold)


@pytest.mark.gpu
def test_hip_reproduces_golden(gold):
    from agile_grasp2_amd import capi
    d = capi.Detector(**gold["params"])
    check(run_backend(d, gold), gold)
    d.close()
