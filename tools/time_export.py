#!/usr/bin/env python3
"""Time the two candidate export forms on the GPU box: python tools/time_export.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import scene_params
from agile_grasp2_amd import capi, scene, sharding
xyz, ws = scene.make_scene(1, 300000, voxel=scene.VOXEL)
idx = scene.draw_samples(1, xyz.shape[0], 5000)
d = capi.Detector(**scene_params(ws))
d.set_stream(torch.cuda.current_stream().cuda_stream)
d.set_cloud(xyz); d.compute_normals()
recs = d.generate_hypotheses(sample_idx=idx, seed=1)
n_slots = 5000 * 8
full = torch.empty(n_slots * 176, dtype=torch.uint8, device="cuda")
cap = 2 * len(recs)
comp = torch.empty(sharding.compact_bytes(cap), dtype=torch.uint8, device="cuda")
def t(fn, reps=200):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
print("records", len(recs), "cap", cap)
print("full table export     %.1f us" % t(lambda: d.export_candidates_device(full.data_ptr(), full.numel())))
print("compact export        %.1f us" % t(lambda: d.export_candidates_compact_device(comp.data_ptr(), comp.numel(), cap)))
