import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu() -> bool:
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


def pytest_collection_modifyitems(config, items):
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (no /dev/kfd)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


from agile_grasp2_amd import scene  # noqa: E402

PARAMS = dict(init_bite=0.01, num_orientations=8, min_score_diff=0.0, num_selected=30,
              filter_half_grasps=0, min_aperture=0.03, max_aperture=0.08)


def scene_params(ws, **kw):
    d = dict(PARAMS)
    d.update(cam_origin=[scene.CAMERA, scene.CAMERA], workspace=list(ws))
    d.update(kw)
    return d


@pytest.fixture(scope="session")
def small_scene():
    xyz, ws = scene.make_scene(seed=3, n_target=6000)
    idx = scene.draw_samples(3, xyz.shape[0], 120)
    return xyz, ws, idx


@pytest.fixture(scope="session")
def oracle_small(small_scene):
    from oracle import api
    xyz, ws, idx = small_scene
    o = api.Oracle(**scene_params(ws, num_threads=4))
    o.set_cloud(xyz)
    o.compute_normals()
    return o


def np_rng(seed):
    return np.random.default_rng(seed)
