#!/usr/bin/env python3
"""Extra seeds of the randomised parity sweep (tests/test_gpu_fuzz.py) plus denser scenes that reach
the sweep's global-slice and global-scratch paths: python tools/fuzz_more.py [first] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_fuzz as tf
from conftest import scene_params
from agile_grasp2_amd import capi, scene
from oracle import api

first = int(sys.argv[1]) if len(sys.argv) > 1 else 24
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for seed in range(first, first + count):
    tf.test_random_configuration_matches_oracle(seed)
    print("fuzz seed", seed, "ok", flush=True)
# dense, un-voxelised scenes: long cropped lists (global slice of stage 0, stage 1), long pieces
for seed, n, kind, R in ((1, 150000, "objects", 8), (2, 250000, "tabletop", 16), (3, 320000, "tabletop", 12)):
    xyz, ws = scene.make_scene(seed=seed, n_target=n, kind=kind, voxel=None)
    idx = scene.draw_samples(seed, xyz.shape[0], 120)
    prm = scene_params(ws, num_threads=16, num_orientations=R)
    d, o = capi.Detector(**prm), api.Oracle(**prm)
    for x in (d, o):
        x.set_cloud(xyz)
        x.compute_normals()
    hd = d.generate_hypotheses(sample_idx=idx, seed=seed)
    ho = o.generate_hypotheses(sample_idx=idx, seed=seed)
    c = d.counters()
    assert hd.tobytes() == ho.tobytes(), seed
    assert np.array_equal(d.prune(len(hd)), o.prune(len(ho)))
    k = min(len(ho), 30)
    if k:
        assert np.array_equal(d.render_images(0, k), o.render_images(0, k))
    print("dense", n, kind, "hyps", len(hd), "mean Kcrop", c.sum_kcrop // max(1, c.n_frames), "handed on", c.n_overflow_samples, "ok", flush=True)
    d.close()
print("all ok")
