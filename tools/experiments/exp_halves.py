"""Feasibility of splitting one detect step into sample halves that overlap on two HIP streams: the headline cloud
through ag2_pipe with its samples in 1, 2 and 4 parts (every part repeats grid + normals: an upper bound of the
cost, a lower bound of the gain).  Prints ms per whole cloud."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from agile_grasp2_amd import capi, scene
from agile_grasp2_amd.weights import make_lenet_weights

n_points, S, R, _, kind = bench.CONFIGS["cfg2"]
xyz, ws = scene.make_scene(1, n_points, kind=kind, voxel=scene.VOXEL)
rng = np.random.default_rng(1)
idx = np.sort(rng.choice(xyz.shape[0], S, replace=False)).astype(np.int32)
weights = make_lenet_weights(7)
xyz_dev = torch.from_numpy(xyz).cuda()
for parts in (1, 2, 4):
    pipe = capi.Pipe(device=0, depth=2, **bench.launch_params(ws, R))
    pipe.lenet_load(weights)
    cuts = [idx[k * S // parts:(k + 1) * S // parts].copy() for k in range(parts)]
    def cloud():
        sc = 0
        pend = 0
        for part in cuts:
            pipe.submit(sample_idx=part, seed=1, dptr=xyz_dev.data_ptr(), n=xyz.shape[0], stride=12)
            pend += 1
            if pend == 2:
                sc += pipe.wait()[1]
                pend -= 1
        while pend:
            sc += pipe.wait()[1]
            pend -= 1
        return sc
    for _ in range(6):
        cloud()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 40
    for _ in range(reps):
        sc = cloud()
    dt = (time.perf_counter() - t0) / reps
    print(f"parts {parts}: {dt*1e3:.3f} ms per cloud, scored {sc}", flush=True)
    pipe.close()
