"""CPU-side checks of the product boundary (no GPU, no compute calls):
the C-ABI library builds, loads and exports every symbol include/ag2_c.h declares; struct layouts
seen by the harness match the header; creating a context without a GPU fails loudly (there is no
CPU fallback); the oracle is not linked into the product."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import __graft_entry__ as entry
from conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "agile_grasp2_amd", "csrc", "libag2hip.so")


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(LIB):
        entry.build()
    return LIB


def test_library_exports_every_declared_symbol(built):
    declared = entry.declared_symbols()
    assert len(declared) >= 20 and "ag2_detect" in declared and "ag2_set_cloud" in declared
    exported = entry.exported_symbols(built)
    missing = [s for s in declared if s not in exported]
    assert not missing, missing
    from agile_grasp2_amd import capi
    assert sorted(capi.SYMBOLS) == declared  # the harness binding covers the whole ABI


def test_library_loads_and_reports_abi_version(built):
    lib = ctypes.CDLL(built)
    assert lib.ag2_abi_version() == 1


def test_struct_layouts_match_header(built):
    from agile_grasp2_amd import capi
    from oracle import api
    assert capi.HYP_DTYPE.itemsize == 176 and api.HYP_DTYPE.itemsize == 176
    assert ctypes.sizeof(capi.Params) == ctypes.sizeof(api.Params)
    # ag2_params: 9 doubles, 4 int32, 2x3 + 6 doubles, 3 doubles, 2 int32
    assert ctypes.sizeof(capi.Params) == 9 * 8 + 4 * 4 + 6 * 8 + 6 * 8 + 3 * 8 + 2 * 4
    assert ctypes.sizeof(capi.Counters) == 16 * 8
    assert ctypes.sizeof(capi.Times) == 12 * 4
    p = capi.default_params()
    assert p.finger_width == 0.01 and p.hand_outer_diameter == 0.09 and p.num_orientations == 8
    assert p.init_bite == 0.015 and p.min_score_diff == 500.0 and p.num_selected == 50  # grasp_detector.cpp:19-80
    text = open(os.path.join(ROOT, "include", "ag2_c.h")).read()
    for field, _ in capi.Params._fields_:
        assert re.search(r"\b%s\b" % field, text), field


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a machine without a GPU")
def test_no_silent_cpu_fallback(built):
    from agile_grasp2_amd import capi
    with pytest.raises(RuntimeError):
        capi.Detector()


def test_product_does_not_link_the_oracle(built):
    out = subprocess.check_output(["readelf", "-d", built], text=True)
    assert "ag2_oracle" not in out
    syms = entry.exported_symbols(built)
    assert not any(s.startswith("ag2o_") for s in syms)
    for root, _, files in os.walk(os.path.join(ROOT, "agile_grasp2_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(root, f)).read()
                assert "from oracle" not in text and "import oracle" not in text, f
                assert "ag2_oracle.h" not in text, f


def test_scene_generator_voxel_semantics():
    """CloudCamera::voxelizeCloud (cloud_camera.cpp:124-168): corner value floor((p-min)/c)*c+min,
    unique, lexicographic (ix,iy,iz) order."""
    from agile_grasp2_amd import scene
    rng = np.random.default_rng(0)
    p = rng.uniform(0, 0.05, size=(5000, 3)).astype(np.float32)
    v = scene.voxelize(p, 0.003)
    mn = p.min(axis=0)
    ijk = np.rint((v - mn) / np.float32(0.003)).astype(np.int64)
    assert len(np.unique(ijk, axis=0)) == len(ijk)
    key = (ijk[:, 0] * 100 + ijk[:, 1]) * 100 + ijk[:, 2]
    assert np.all(np.diff(key) > 0)
    assert np.all(v >= mn - 1e-6) and np.all(v <= p.max(axis=0) + 1e-6)
    xyz, ws = scene.make_scene(1, 4000)
    assert abs(xyz.shape[0] - 4000) < 0.05 * 4000 and xyz.dtype == np.float32
    idx = scene.draw_samples(1, xyz.shape[0], 100)
    assert len(np.unique(idx)) == 100 and np.all(np.diff(idx) > 0)
