#!/usr/bin/env python3
"""Generate tests/golden/golden_small.npz from the CPU oracle.

The reference (gwding/agile_grasp2) holds no fixtures, golden vectors or known-answer tests for
this path and cannot be built or run in this image, so these vectors are produced by the oracle
(oracle/ag2_oracle.cpp, a restatement of the reference algorithm, itself pinned to independent
numpy / scipy / torch re-derivations in tests/test_oracle_pins.py).  They are data only: seeded
synthetic inputs and the outputs every later build must reproduce.  Regenerate with

    python tests/golden/make_golden.py

Contents (all little-endian): xyz [N,3] f32, sample_idx [S] i32, params (json), seed, slot_base,
normals [3,N] f32 bits, frames [S,12] f64 + valid [S], hypothesis records (176-byte structured),
prune flags, sha256 of every 60x60x3 image + the first 6 images in full, in-box point lists of the
first 3 hypotheses, LeNet weights seed, logits of all images, detect() outputs.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from agile_grasp2_amd import scene  # noqa: E402
from agile_grasp2_amd.weights import make_lenet_weights  # noqa: E402
from oracle import api  # noqa: E402

SCENE_SEED, N_TARGET, N_SAMPLES, SEED, SLOT_BASE, WEIGHT_SEED = 3, 6000, 120, 5, 40, 7


def golden_params(ws):
    return dict(init_bite=0.01, num_orientations=8, min_score_diff=-1e30, num_selected=25,
                filter_half_grasps=0, min_aperture=0.03, max_aperture=0.08,
                cam_origin=[[float(v) for v in scene.CAMERA]] * 2, workspace=[float(v) for v in ws])


def main():
    xyz, ws = scene.make_scene(seed=SCENE_SEED, n_target=N_TARGET)
    idx = scene.draw_samples(SCENE_SEED, xyz.shape[0], N_SAMPLES)
    prm = golden_params(ws)
    o = api.Oracle(**prm)
    o.set_cloud(xyz)
    o.compute_normals()
    normals = o.get_normals().astype(np.float32)
    frames, valid = o.local_frames(sample_idx=idx, slot_base=SLOT_BASE, seed=SEED)
    hyps = o.generate_hypotheses(sample_idx=idx, slot_base=SLOT_BASE, seed=SEED)
    keep = o.prune(len(hyps))
    imgs = o.render_images(0, len(hyps))
    sha = np.array([hashlib.sha256(im.tobytes()).hexdigest() for im in imgs])
    lists = [o.hyp_points(k, int(hyps[k]["n_points"])) for k in range(3)]
    w = make_lenet_weights(WEIGHT_SEED)
    o.lenet_load(w)
    logits = o.lenet_forward(imgs)
    sel, scored = o.detect(sample_idx=idx, slot_base=SLOT_BASE, seed=SEED, do_prune=True)
    out = os.path.join(HERE, "golden_small.npz")
    np.savez_compressed(
        out, xyz=xyz, sample_idx=idx, params=json.dumps(prm), seed=SEED, slot_base=SLOT_BASE,
        weight_seed=WEIGHT_SEED, normals_bits=normals.view(np.uint32), frames=frames, frames_valid=valid,
        hyps=hyps, prune_keep=keep, image_sha256=sha, images_first=imgs[:6],
        list0_pts=lists[0][0], list0_nrm=lists[0][1], list1_pts=lists[1][0], list1_nrm=lists[1][1],
        list2_pts=lists[2][0], list2_nrm=lists[2][1], logits=logits, selected=sel, scored=scored)
    print(f"wrote {out}: {os.path.getsize(out)} bytes, {len(hyps)} hypotheses, {len(scored)} scored, "
          f"{len(sel)} selected")


if __name__ == "__main__":
    main()
