#!/usr/bin/env python3
"""Time detect + exchange (world 1 rehearsal) with both export forms: python tools/time_exchange.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist
from conftest import scene_params
from agile_grasp2_amd import capi, scene, sharding
from agile_grasp2_amd.weights import make_lenet_weights
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
xyz, ws = scene.make_scene(1, 300000, voxel=scene.VOXEL)
idx = scene.draw_samples(1, xyz.shape[0], 5000)
d = capi.Detector(**scene_params(ws, min_score_diff=300.0))
d.set_stream(torch.cuda.current_stream().cuda_stream)
d.set_stage_timing(1)
d.lenet_load(make_lenet_weights(7))
xd = torch.from_numpy(xyz).cuda()
full = torch.empty(40000 * 176, dtype=torch.uint8, device="cuda")
cap = 4724
comp = torch.empty(sharding.compact_bytes(cap), dtype=torch.uint8, device="cuda")
def base():
    d.set_cloud_device(xd.data_ptr(), xyz.shape[0], 12); d.compute_normals()
    d.detect(sample_idx=idx, seed=1, want_all=False)
def with_full():
    base(); d.export_candidates_device(full.data_ptr(), full.numel()); sharding.all_gather_tables(full, 1)
def with_compact():
    base(); d.export_candidates_compact_device(comp.data_ptr(), comp.numel(), cap); sharding.all_gather_tables(comp, 1)
def only_export_compact():
    base(); d.export_candidates_compact_device(comp.data_ptr(), comp.numel(), cap)
def t(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for name, fn in (("detect only", base), ("+ full export + all_gather", with_full), ("+ compact export + all_gather", with_compact), ("+ compact export only", only_export_compact), ("detect only", base)):
    print("%-34s %.4f ms" % (name, t(fn)))
dist.destroy_process_group()
