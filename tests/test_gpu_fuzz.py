"""Randomised parity sweep: small scenes, random hand geometry / radii / orientation counts (including
geometries for which the kernels' fast float32 classification is switched off and every decision takes
the exact f64 path), HIP path vs oracle.  Records, prune flags, image bytes: bit-exact; scores within
the LeNet tolerance.  Seeds are fixed: a failure reproduces."""
import numpy as np
import pytest

from conftest import scene_params
from agile_grasp2_amd import scene
from agile_grasp2_amd.weights import make_lenet_weights

pytestmark = pytest.mark.gpu


def draw_case(seed):
    rng = np.random.default_rng(1000 + seed)
    prm = dict(
        num_orientations=int(rng.choice([4, 8, 12, 16])),
        finger_width=float(rng.choice([0.008, 0.01, 0.012])),
        hand_outer_diameter=float(rng.choice([0.08, 0.09, 0.10])),
        hand_depth=float(rng.choice([0.05, 0.06, 0.07])),
        hand_height=float(rng.choice([0.015, 0.02])),
        init_bite=float(rng.choice([0.01, 0.015])),
        nn_radius_hands=float(rng.choice([0.06, 0.08, 0.10])),
        nn_radius_taubin=float(rng.choice([0.01, 0.012])),
        filter_half_grasps=int(rng.integers(0, 2)),
        min_score_diff=-1e30, num_selected=100000,
    )
    kind = str(rng.choice(["tabletop", "objects"]))
    n = int(rng.choice([2500, 5000, 9000]))
    return prm, kind, n, int(rng.integers(60, 160))


@pytest.mark.parametrize("seed", range(24))
def test_random_configuration_matches_oracle(seed):
    from agile_grasp2_amd import capi
    from oracle import api
    prm, kind, n, n_samples = draw_case(seed)
    xyz, ws = scene.make_scene(seed=50 + seed, n_target=n, kind=kind)
    idx = scene.draw_samples(seed, xyz.shape[0], n_samples)
    full = scene_params(ws, num_threads=4, **prm)
    d, o = capi.Detector(**full), api.Oracle(**full)
    w = make_lenet_weights(seed)
    for x in (d, o):
        x.set_cloud(xyz)
        x.compute_normals()
        x.lenet_load(w)
    assert np.array_equal(d.get_normals(), o.get_normals(), equal_nan=True)
    hd = d.generate_hypotheses(sample_idx=idx, seed=seed)
    ho = o.generate_hypotheses(sample_idx=idx, seed=seed)
    assert len(hd) == len(ho)
    assert hd.tobytes() == ho.tobytes()
    if len(ho):
        assert np.array_equal(d.prune(len(hd)), o.prune(len(ho)))
        k = min(len(ho), 40)
        assert np.array_equal(d.render_images(0, k), o.render_images(0, k))
        for h in range(0, len(ho), max(1, len(ho) // 6)):
            pd_, nd_ = d.hyp_points(h, int(hd[h]["n_points"]))
            po_, no_ = o.hyp_points(h, int(ho[h]["n_points"]))
            assert pd_.tobytes() == po_.tobytes() and nd_.tobytes() == no_.tobytes()
    sel_d, all_d = d.detect(sample_idx=idx, seed=seed)
    sel_o, all_o = o.detect(sample_idx=idx, seed=seed)
    assert len(all_d) == len(all_o) and len(sel_d) == len(sel_o)
    if len(all_o):
        for f in ("sample_slot", "orientation", "half_antipodal", "n_points", "width", "bottom", "top",
                  "surface", "axis", "approach", "binormal"):
            assert np.array_equal(all_d[f], all_o[f]), f
        tol = 1e-4 * np.abs(all_o["score"]).max() + 2e-3
        assert np.abs(all_d["score"] - all_o["score"]).max() <= tol
    d.close()
