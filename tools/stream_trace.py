#!/usr/bin/env python3
"""Per-frame latencies of the cfg5 stream's step-by-step leg (diagnostic): python tools/stream_trace.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from agile_grasp2_amd import capi, scene
from agile_grasp2_amd.weights import make_lenet_weights

n_points, S, R, _, _ = bench.CONFIGS["cfg5"]
raws, ws = scene.make_stream(1, int(2.55 * n_points), 23, voxel=None)
dev = [torch.from_numpy(c).cuda() for c in raws]
torch.cuda.synchronize()
d = capi.Detector(**bench.launch_params(ws, R))
d.lenet_load(make_lenet_weights(7))
d.set_stage_timing(0)
lat = []
for k in range(len(raws)):
    t0 = time.perf_counter()
    vox = d.preprocess_cloud_device(dev[k].data_ptr(), raws[k].shape[0], 12, voxel_size=scene.VOXEL)
    t1 = time.perf_counter()
    ns = d.subsample_uniformly(S, seed=1 + k, want_indices=False)
    t2 = time.perf_counter()
    d.compute_normals()
    sel, n_sc = d.detect(n_resident=ns, seed=1, do_prune=True, want_all=False)
    t3 = time.perf_counter()
    c = d.counters()
    lat.append(((t3 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, int(c.n_overflow_samples), int(c.detect_redone)))
for k, l in enumerate(lat):
    print(k, " ".join(f"{x:.3f}" if isinstance(x, float) else str(x) for x in l))
