"""The A/B switches of the library ("A/B switches") select kernels the default path does
not launch -- the sorting-network renderer for 1 025 .. 4 096-point images, the static deal of the larger
images, the rank / emit pair of the sub-sampling, the general front end, the step-by-step detect.  The switches
are read once per process, so each variant runs a few of the suite's parity tests in a process of its own
(one after the other: one GPU process at a time besides this one)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {
    "AG2_RENDER_BITONIC": ["tests/test_gpu_hypotheses.py::test_images_from_points_edge_cases",
                           "tests/test_gpu_hypotheses.py::test_dense_unvoxelised_cloud_overflow_path"],
    "AG2_RENDER_STATIC": ["tests/test_gpu_hypotheses.py::test_images_from_points_edge_cases",
                          "tests/test_gpu_frames.py::test_frames_of_a_dense_cloud_use_the_renderers_for_large_images"],
    "AG2_SEL_PAIR": ["tests/test_gpu_preprocess.py::test_subsample_matches_oracle",
                     "tests/test_gpu_frames.py::test_raw_frames_equal_the_oracle_and_the_stepwise_path"],
    "AG2_PRE_GENERAL": ["tests/test_gpu_preprocess.py::test_filter_and_voxel_grid_match_oracle",
                        "tests/test_gpu_preprocess.py::test_whole_front_end_then_detect"],
    "AG2_DETECT_STEPWISE": ["tests/test_gpu_lenet_detect.py::test_detect_threshold_and_topk",
                            "tests/test_gpu_lenet_detect.py::test_one_round_trip_detect_equals_the_step_by_step_form"],
    # (the library polls the flags its kernels write behind their results; 0: it waits for the stream instead)
    "AG2_POLL=0": ["tests/test_gpu_lenet_detect.py::test_one_round_trip_detect_equals_the_step_by_step_form",
                   "tests/test_gpu_frames.py::test_frames_equal_the_stepwise_path"],
}


@pytest.mark.parametrize("switch", sorted(CASES))
def test_parity_under_switch(switch):
    name, _, value = switch.partition("=")
    env = dict(os.environ, **{name: value or "1"})
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + CASES[switch],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
