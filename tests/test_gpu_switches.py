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


def test_bench_two_rank_rehearsal_validates_its_merge_against_the_unsplit_run():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per process), rehearsed on the
    one GPU of the box (AG2_BENCH_REHEARSAL=1: both ranks on GPU 0, the all-gather over gloo -- RCCL refuses two
    ranks on one device): spatial tiles, compact exchange, merge on every rank, and then rank 0's self-validation
    (bench.validate_against_unsplit): the merged top-k against ONE unsplit run over all the ranks' samples --
    records byte-equal apart from the score, scores within tolerance, selection and counts consistent
    (hand_search.cpp:194-228, grasp_detector.cpp:228-252).  What the first N-GPU hardware record will carry."""
    import json
    env = dict(os.environ, AG2_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--no-cpu"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    v = out["config"]["per_rank"]["matches_unsplit"]
    assert v["ok"] and v["records_byte_equal"] and v["scores_within_tol"] and v["selection_ok"] and v["counts_ok"], v
    assert v["merged_selected"] == 30 and v["unsplit_scored"] == v["ranks_scored"] > 1000
    assert out["config"]["per_rank"]["halo_duplication"] < 1.3
