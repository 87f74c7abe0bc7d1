"""ag2_pipe with clouds in pageable host memory: ms per cloud (headline cloud 3.6 MB, raw cloud 9.2 MB).
Run with AG2_STAGE_MIN_MB=1000 (the caller copies alone) and without (helper threads) on the same box."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from agile_grasp2_amd import capi, scene
from agile_grasp2_amd.weights import make_lenet_weights
n_points, S, R, _, kind = bench.CONFIGS["cfg2"]
xyz, ws = scene.make_scene(1, n_points, kind=kind, voxel=scene.VOXEL)
idx = scene.draw_samples(1, xyz.shape[0], S)
inside = np.all((xyz > np.array(ws)[[0, 2, 4]]) & (xyz < np.array(ws)[[1, 3, 5]]), axis=1)
raw, _ = scene.raw_from_voxels(xyz[inside], 1), ws
pipe = capi.Pipe(device=0, depth=2, **bench.launch_params(ws, R))
pipe.lenet_load(make_lenet_weights(7))
def run(sub, reps):
    sub(); sc = 0
    for _ in range(reps - 1):
        sub(); sc += pipe.wait()[1]
    return sc + pipe.wait()[1]
for name, sub in (("headline_from_host", lambda: pipe.submit(xyz, idx, seed=1)),
                  ("raw_from_host", lambda: pipe.submit_raw(raw, num_samples=S, sample_seed=1, seed=1, voxel_size=scene.VOXEL))):
    run(sub, 8); torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); run(sub, 40); dt = (time.perf_counter() - t0) / 40
        print(name, "%.3f ms per cloud" % (dt * 1e3), flush=True)
pipe.close()
