#!/usr/bin/env python3
"""The `with_front_end` leg of bench.py on its own (for a profiler): the headline cloud's raw version
(2.55 points per voxel, shuffled) through ag2_detect_frame_raw, `steps` frames after 3 warm-up frames, and
through the separate calls.  Prints one JSON line with the host's view: ms per frame, and -- when the
kernel trace is summarised afterwards -- what is left between the kernels' sum and the wall time.
    python tools/front_end_leg.py [steps]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from agile_grasp2_amd import capi, scene  # noqa: E402
from agile_grasp2_amd.weights import make_lenet_weights  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n_points, S, R, _, kind = bench.CONFIGS["cfg2"]
xyz, ws = scene.make_scene(1, n_points, kind=kind, voxel=scene.VOXEL)
xd = xyz.astype(np.float64)
inside = ((xd[:, 0] > ws[0]) & (xd[:, 0] < ws[1]) & (xd[:, 1] > ws[2]) & (xd[:, 1] < ws[3]) &
          (xd[:, 2] > ws[4]) & (xd[:, 2] < ws[5]))
raw = scene.raw_from_voxels(xyz[inside], 1)
d = capi.Detector(**bench.launch_params(ws, R))
d.lenet_load(make_lenet_weights(7))
d.set_stage_timing(0)
raw_dev = torch.from_numpy(raw).cuda()
torch.cuda.synchronize()


def frame():
    return d.detect_frame_raw(num_samples=S, sample_seed=1, seed=1, do_prune=True, dptr=raw_dev.data_ptr(),
                              n=raw.shape[0], stride=12, voxel_size=scene.VOXEL)


for _ in range(3):
    frame()
torch.cuda.synchronize()
lat = []
for _ in range(steps):
    t0 = time.perf_counter()
    sel, n_sc, n_vox = frame()
    lat.append((time.perf_counter() - t0) * 1e3)
fi = d.frame_info()
print(json.dumps({"leg": "with_front_end (ag2_detect_frame_raw, graph)", "frames": steps, "raw_points": int(raw.shape[0]),
                  "voxels": int(n_vox), "scored_per_frame": int(n_sc), "ms_per_frame_mean": float(np.mean(lat)),
                  "ms_per_frame_p50": float(np.percentile(lat, 50)), "graph_replays": int(fi.graph_replays),
                  "fallbacks": int(fi.fallbacks)}))
