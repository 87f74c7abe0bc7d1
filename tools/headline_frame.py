#!/usr/bin/env python3
"""The headline workload (cfg2) through the three synchronous entries, same box, same process:
set_cloud_device + compute_normals + detect, detect_frame with the graph, detect_frame without."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from agile_grasp2_amd import capi, scene
from agile_grasp2_amd.weights import make_lenet_weights

n_points, S, R, _, kind = bench.CONFIGS["cfg2"]
xyz, ws = scene.make_scene(1, n_points, kind=kind, voxel=scene.VOXEL)
idx = scene.draw_samples(1, xyz.shape[0], S)
w = make_lenet_weights(7)
dev = torch.from_numpy(xyz).cuda()
torch.cuda.synchronize()
out = {}
legs = sys.argv[1:] or ["stepwise", "frame_graph", "frame_plain"]
for name in legs:
    d = capi.Detector(**bench.launch_params(ws, R))
    d.lenet_load(w)
    d.set_stage_timing(int(os.environ.get("TIMING", "0")))
    if name != "stepwise":
        d.stream_configure(0, 0, name == "frame_graph")

    def step():
        if name == "stepwise":
            d.set_cloud_device(dev.data_ptr(), xyz.shape[0], 12)
            d.compute_normals()
            return d.detect(sample_idx=idx, seed=1, do_prune=True, want_all=False)[1]
        return d.detect_frame(sample_idx=idx, seed=1, do_prune=True, dptr=dev.data_ptr(), n=xyz.shape[0], stride=12)[1]

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for _ in range(40):
        n += step()
    torch.cuda.synchronize()
    out[name] = {"ms_per_step": (time.perf_counter() - t0) / 40 * 1e3, "scored": n // 40}
    d.close()
print(json.dumps(out))
