# experiment: per-unit phase stamps of k_lenet_conv_x3b (build with -DAG2_EXP_DBG=1, AG2_LIB=that build)
import ctypes, os, sys, json, subprocess
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from agile_grasp2_amd import capi, scene
from agile_grasp2_amd.weights import make_lenet_weights
n_points, S, R, vox, kind = bench.CONFIGS["cfg2"]
xyz, ws = scene.make_scene(1, n_points, kind=kind, voxel=scene.VOXEL)
idx = scene.draw_samples(1, xyz.shape[0], S)
d = capi.Detector(**bench.launch_params(ws, R))
d.set_cloud(xyz); d.compute_normals(); d.lenet_load(make_lenet_weights(7))
for _ in range(3): d.detect(sample_idx=idx, seed=1, do_prune=True)
lib = ctypes.CDLL(os.environ["AG2_LIB"])
buf = np.zeros(512 * 64 * 4, dtype=np.uint64)
lib.ag2_dbg_x3(buf.ctypes.data_as(ctypes.c_void_p))
a = buf.reshape(512, 4, 8, 8)   # wg, wave, round, stamp
ok = a[:, :, :5, 0] > 0
t = a.astype(np.int64)
print("hw_id samples (wg, cu bits):", [(w, hex(int(a[w,0,0,7]))) for w in (0,1,8,256,257,264)])
for rnd in range(5):
    m = t[:, 0, rnd, 0] > 0
    st = t[m][:, 0, rnd, :5]
    dlt = np.diff(st, axis=1)
    print("round", rnd, "n", m.sum(), "put+bar %.0f conv1 %.0f bar2 %.0f conv2 %.0f" % tuple(np.median(dlt, axis=0)), " total %.0f" % np.median(st[:,4]-st[:,0]))
# clock: cycles vs wall clock between rounds 0 and 4
m = (t[:,0,4,0] > 0)
cyc = (t[m][:,0,4,0] - t[m][:,0,0,0]); wall = (t[m][:,0,4,6] - t[m][:,0,0,6])
print("cycle counter ticks per 100MHz tick: %.2f" % np.median(cyc / np.maximum(wall,1)))
# offset between wg i and wg i+256 at round 1 start
off = t[256:512,0,1,0] - t[0:256,0,1,0]
print("start offset wg i+256 vs wg i (round 1): median %.0f, abs median %.0f" % (np.median(off), np.median(np.abs(off))))
buf2 = np.zeros(512 * 4 * 32, dtype=np.uint64)
lib.ag2_dbg_x3b(buf2.ctypes.data_as(ctypes.c_void_p))
b2 = buf2.reshape(512, 4, 32).astype(np.int64)[:, :, :16]
dd = np.diff(b2, axis=2)          # cycles per pair of blocks
print("conv2 cycles per pair of blocks (round 2), median over waves, per iteration:")
print(np.median(dd.reshape(-1, 15), axis=0).astype(int))
print("p10/p90 of all:", np.percentile(dd, 10), np.percentile(dd, 90))
print("conv2 start..first stamp vs phase stamp3: ", np.median(b2[:,0,0] - t[:,0,2,3]))
