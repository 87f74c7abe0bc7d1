import sys; sys.path.insert(0, ".")
import numpy as np, bench
from agile_grasp2_amd import capi, scene
n_points, S, R, vox, kind = bench.CONFIGS["cfg3"]
xyz, ws = scene.make_scene(1, n_points, kind=kind, voxel=None)
idx = scene.draw_samples(1, xyz.shape[0], S)
d = capi.Detector(**bench.launch_params(ws, R))
d.set_cloud(xyz); d.compute_normals()
h = d.generate_hypotheses(sample_idx=idx, seed=1)
keep = d.prune(len(h)).astype(bool)
p = h["n_points"][keep]
print("images", len(p), "mean", p.mean(), "max", p.max())
for lo, hi in ((0,1024),(1024,2048),(2048,4096),(4096,8192),(8192,16384),(16384,1<<30)):
    sel = (p > lo) & (p <= hi)
    print(lo, hi, int(sel.sum()), int(p[sel].sum()))
