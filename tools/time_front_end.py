#!/usr/bin/env python3
"""Time the GPU front end (workspace filter + voxel grid + sub-sampling) on a raw cloud that is
already in HBM.  Run on the GPU box: python tools/time_front_end.py [n_raw]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import scene_params  # noqa: E402
from agile_grasp2_amd import capi, scene  # noqa: E402

n_raw = int(sys.argv[1]) if len(sys.argv) > 1 else 1300000
raw, ws = scene.make_scene(seed=1, n_target=n_raw, voxel=None, spacing=0.0015)
d = capi.Detector(**scene_params(ws))
hip = C.CDLL("libamdhip64.so.7")
dptr = C.c_void_p()
assert hip.hipMalloc(C.byref(dptr), C.c_size_t(raw.nbytes)) == 0
assert hip.hipMemcpy(dptr, raw.ctypes.data_as(C.c_void_p), C.c_size_t(raw.nbytes), C.c_int(1)) == 0
for it in range(6):
    t0 = time.perf_counter()
    m = d.preprocess_cloud_device(dptr.value, raw.shape[0], 12)
    t1 = time.perf_counter()
    k = d.subsample_uniformly(5000, seed=it, want_indices=False)
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    t = d.times()
    print(f"run {it}: {raw.shape[0]} raw -> {m} voxels, {k} samples | preprocess wall {1e3 * (t1 - t0):.3f} ms "
          f"(device {t.preprocess_ms:.3f} ms + grid {t.grid_ms:.3f} ms) | subsample wall {1e3 * (t2 - t1):.3f} ms")
