"""ag2_pipe throughput against its depth (frames in flight), headline cloud resident in HBM."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from agile_grasp2_amd import capi, scene
from agile_grasp2_amd.weights import make_lenet_weights
n_points, S, R, _, kind = bench.CONFIGS["cfg2"]
xyz, ws = scene.make_scene(1, n_points, kind=kind, voxel=scene.VOXEL)
idx = scene.draw_samples(1, xyz.shape[0], S)
dev = torch.from_numpy(xyz).cuda()
w = make_lenet_weights(7)
for depth in (1, 2, 3, 4):
    pipe = capi.Pipe(device=0, depth=depth, **bench.launch_params(ws, R))
    pipe.lenet_load(w)
    def run(reps):
        pend = 0; sc = 0
        for _ in range(reps):
            if pend == depth:
                sc += pipe.wait()[1]; pend -= 1
            pipe.submit(sample_idx=idx, seed=1, dptr=dev.data_ptr(), n=xyz.shape[0], stride=12); pend += 1
        while pend:
            sc += pipe.wait()[1]; pend -= 1
        return sc
    run(4 * depth + 4); torch.cuda.synchronize()
    for rep in range(4):
        t0 = time.perf_counter(); run(60); dt = (time.perf_counter() - t0) / 60
        print("depth", depth, "%.4f ms per cloud" % (dt * 1e3), flush=True)
    pipe.close()
