"""Invariants of the reference algorithm (each derived from the cited code),
checked on the oracle's output -- and, on the GPU box, on the HIP path at the full BASELINE
config-2 size where the oracle is too slow to be the checker (size-independent properties).
"""
import numpy as np
import pytest

from conftest import scene_params
from agile_grasp2_amd import scene


def check_invariants(hyps, prm, xyz=None):
    od, fw, depth = prm["hand_outer_diameter"], prm["finger_width"], prm["hand_depth"]
    B, A, X = hyps["binormal"], hyps["approach"], hyps["axis"]
    # hand_search.cpp:383-385: columns of frame_rot -> orthonormal, right-handed (binormal x approach = axis)
    for V in (B, A, X):
        assert np.allclose(np.linalg.norm(V, axis=1), 1.0, atol=1e-9)
    assert np.allclose(np.einsum("ij,ij->i", B, A), 0.0, atol=1e-9)
    assert np.allclose(np.einsum("ij,ij->i", B, X), 0.0, atol=1e-9)
    assert np.allclose(np.cross(B, A), X, atol=1e-9)
    # finger_hand.cpp:194-199: bottom = top - hand_depth * approach ; surface on the same line
    assert np.allclose(hyps["top"] - hyps["bottom"], depth * A, atol=1e-9)
    d_surf = np.einsum("ij,ij->i", hyps["surface"] - hyps["bottom"], A)
    assert np.allclose(hyps["surface"] - hyps["bottom"], d_surf[:, None] * A, atol=1e-9)
    # hand_search.cpp:397 + finger_hand.cpp:155-156: 0 <= width <= right - left = od - 2 fw
    assert np.all(hyps["width"] >= 0) and np.all(hyps["width"] <= od - 2 * fw + 1e-12)
    # hand_search.cpp:417-418: FULL => HALF
    assert np.all(hyps["half_antipodal"] >= hyps["full_antipodal"])
    assert np.all(hyps["n_points"] > 0)
    # output order = sample order, then orientation (hand_search.cpp:223-228)
    key = hyps["sample_slot"].astype(np.int64) * 64 + hyps["orientation"]
    assert np.all(np.diff(key) > 0)
    assert hyps["orientation"].min() >= 0 and hyps["orientation"].max() < prm["num_orientations"]


def check_lists(pts, nrm, prm):
    # hand_search.cpp:404-409: scaled x in (0.15, 0.85) for od = 0.09, fw = 0.01; y, z in [0, 1)
    od, fw = prm["hand_outer_diameter"], prm["finger_width"]
    lo = 0.5 - 0.5 * (od - 2 * fw) / 0.1
    assert pts[0].min() > lo - 1e-9 and pts[0].max() < 1 - lo + 1e-9
    assert pts[1].min() >= -1e-9 and pts[1].max() < 1 + 1e-9
    assert pts[2].min() > -1e-9 and pts[2].max() < 1 + 1e-9
    fin = np.isfinite(nrm).all(axis=0)
    assert np.allclose(np.linalg.norm(nrm[:, fin], axis=0), 1.0, atol=1e-5)  # rotated unit normals


PRM = dict(hand_outer_diameter=0.09, finger_width=0.01, hand_depth=0.06, num_orientations=8)


def test_oracle_invariants(small_scene, oracle_small):
    xyz, ws, idx = small_scene
    hyps = oracle_small.generate_hypotheses(sample_idx=idx, seed=2)
    assert len(hyps) > 20
    check_invariants(hyps, PRM)
    for k in range(0, len(hyps), 5):
        pts, nrm = oracle_small.hyp_points(k, int(hyps[k]["n_points"]))
        check_lists(pts, nrm, PRM)
    imgs = oracle_small.render_images(0, len(hyps))
    assert imgs.dtype == np.uint8 and imgs.max() <= 255
    # zero where no point within the 3x3 neighbourhood: columns outside the scaled x-range + 1 pixel
    assert imgs[:, :, :7, :].max() == 0 and imgs[:, :, 53:, :].max() == 0


def test_oracle_invariants_survive_point_permutation():
    """Shuffling the order points are handed over changes the canonical (cell, index) neighbour order,
    hence which <= 50 normals a frame draws (hand_search.cpp:119-135) -- the hypotheses may differ
    a little, but every geometric invariant must hold for both, and the amount of output stays
    comparable.  Same normals are supplied to both so only the ordering differs."""
    from oracle import api
    xyz, ws = scene.make_scene(seed=8, n_target=4000, kind="objects")
    n = xyz.shape[0]
    idx = scene.draw_samples(8, n, 60)
    perm = np.random.default_rng(0).permutation(n)
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    o0 = api.Oracle(**scene_params(ws))
    o0.set_cloud(xyz)
    o0.compute_normals()
    nrm = o0.get_normals()
    o1 = api.Oracle(**scene_params(ws))
    o1.set_cloud(xyz, normals=nrm)
    o2 = api.Oracle(**scene_params(ws))
    o2.set_cloud(xyz[perm], normals=np.ascontiguousarray(nrm[:, perm]))
    h1 = o1.generate_hypotheses(sample_idx=idx, seed=3)
    h2 = o2.generate_hypotheses(sample_idx=inv[idx].astype(np.int32), seed=3)
    check_invariants(h1, PRM)
    check_invariants(h2, PRM)
    assert len(h1) > 10 and abs(len(h1) - len(h2)) <= max(4, len(h1) // 5)


@pytest.mark.gpu
def test_hip_invariants_at_config2_size():
    """BASELINE config 2 (300k-point voxelised cloud, 5000 samples, 8 orientations) through the HIP
    path: size-independent properties + idempotence (a second run returns identical bytes) +
    sharding independence (two halves of the sample list, global slots, equal the single run)."""
    from agile_grasp2_amd import capi
    xyz, ws = scene.make_scene(seed=1, n_target=300000)
    idx = scene.draw_samples(1, xyz.shape[0], 5000)
    d = capi.Detector(**scene_params(ws))
    d.set_cloud(xyz)
    d.compute_normals()
    nrm = d.get_normals()
    fin = np.isfinite(nrm).all(axis=0)
    assert fin.mean() > 0.99
    assert np.allclose(np.linalg.norm(nrm[:, fin], axis=0), 1.0, atol=1e-6)
    assert np.all(np.einsum("ij,ij->j", nrm[:, fin], -xyz[fin].T.astype(np.float64)) >= -1e-6)
    h = d.generate_hypotheses(sample_idx=idx, seed=4)
    assert len(h) > 500
    check_invariants(h, PRM)
    for k in range(0, len(h), 97):
        pts, nr = d.hyp_points(k, int(h[k]["n_points"]))
        check_lists(pts, nr, PRM)
    h_again = d.generate_hypotheses(sample_idx=idx, seed=4)
    assert h.tobytes() == h_again.tobytes()
    a = d.generate_hypotheses(sample_idx=idx[:2500], slot_base=0, seed=4)
    b = d.generate_hypotheses(sample_idx=idx[2500:], slot_base=2500, seed=4)
    assert np.concatenate([a, b]).tobytes() == h.tobytes()
    d.close()


def test_raw_cloud_generators_voxelise_back():
    """The harness's raw-cloud generators (bench.py's front-end leg, the cfg5 stream): raw_from_voxels gives
    a cloud whose voxelisation (CloudCamera::voxelizeCloud, cloud_camera.cpp:124-168) is the voxel cloud it was
    made from, byte for byte -- in numpy and through the oracle's preprocess_cloud --, and make_stream(voxel=None)
    gives raw frames that voxelise to about 0.4 of their size."""
    from oracle import api
    vox, ws = scene.make_scene(seed=5, n_target=20000)
    x = vox.astype(np.float64)
    inside = ((x[:, 0] > ws[0]) & (x[:, 0] < ws[1]) & (x[:, 1] > ws[2]) & (x[:, 1] < ws[3]) &
              (x[:, 2] > ws[4]) & (x[:, 2] < ws[5]))
    assert inside.all()   # (this scene lies inside its workspace: the lattice origin stays where it was)
    raw = scene.raw_from_voxels(vox, 5)
    assert 2.4 * len(vox) < len(raw) < 2.7 * len(vox)
    assert scene.voxelize(raw).tobytes() == vox.tobytes()
    o = api.Oracle(workspace=list(ws), num_threads=2)
    assert o.preprocess_cloud(raw, voxel_size=scene.VOXEL) == len(vox)
    assert o.get_cloud()[0].tobytes() == vox.tobytes()
    frames, _ = scene.make_stream(3, 30000, 2, voxel=None)
    for f in frames:
        assert abs(len(f) - 30000) < 600
        assert 0.3 * len(f) < len(scene.voxelize(f)) < 0.6 * len(f)
