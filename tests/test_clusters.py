"""Grasp clustering (HandleSearch::findClusters, handle_search.cpp:4-80).
not-gpu: the oracle against an independent vectorised numpy statement.
gpu: the HIP path (ag2_find_clusters, and ag2_detect with ag2_set_min_inliers) against the oracle --
every f64 sum is taken in the reference's own order, so records are compared byte for byte."""
import numpy as np
import pytest

from cluster_data import make_hands, numpy_clusters
from conftest import scene_params
from agile_grasp2_amd import scene, weights
from oracle import api

FIELDS = ("axis", "approach", "binormal", "width", "sample_slot", "orientation", "n_points")


@pytest.mark.parametrize("n,min_inliers", [(0, 3), (1, 1), (400, 1), (400, 3), (1500, 5)])
def test_oracle_clusters_match_numpy(n, min_inliers):
    h = make_hands(api.HYP_DTYPE, seed=n + min_inliers, n=n)
    o = api.Oracle()
    got = o.find_clusters(h, min_inliers)
    if n == 0:
        assert len(got) == 0
        return
    want, cnt = numpy_clusters(h, min_inliers)
    assert len(got) == len(want)
    if n >= 400:
        assert 0.05 * n < len(got) < n        # some hands cluster, the outliers do not
    for f in FIELDS:
        assert np.array_equal(got[f], want[f]), f
    for f in ("surface", "bottom", "top"):
        assert np.allclose(got[f], want[f], rtol=0, atol=1e-12), f
    assert np.allclose(got["score"], want["score"], rtol=1e-12, atol=1e-9)


def test_oracle_remove_inliers_is_order_dependent_and_smaller():
    h = make_hands(api.HYP_DTYPE, seed=3, n=600)
    o = api.Oracle()
    a = o.find_clusters(h, 3, remove_inliers=False)
    b = o.find_clusters(h, 3, remove_inliers=True)
    assert 0 < len(b) < len(a)                 # used inliers cannot vote again (:31-32, :58-59)
    assert set(b["sample_slot"]) <= set(a["sample_slot"])


@pytest.mark.gpu
@pytest.mark.parametrize("n,min_inliers", [(0, 2), (1, 1), (255, 1), (256, 3), (257, 3), (3000, 5), (20000, 8)])
def test_hip_clusters_equal_oracle(n, min_inliers):
    from agile_grasp2_amd import capi
    h = make_hands(capi.HYP_DTYPE, seed=n, n=n, n_lines=max(4, n // 40))
    d = capi.Detector()
    got = d.find_clusters(h, min_inliers)
    want = api.Oracle().find_clusters(h, min_inliers)
    assert len(got) == len(want)
    assert got.tobytes() == want.tobytes()
    if n >= 3000:
        assert 0.02 * n < len(got) < n
    with pytest.raises(RuntimeError):
        d.find_clusters(h, 0)
    d.close()


@pytest.mark.gpu
def test_detect_with_min_inliers_equals_oracle():
    from agile_grasp2_amd import capi
    xyz, ws = scene.make_scene(seed=21, n_target=30000, kind="objects")
    idx = scene.draw_samples(21, xyz.shape[0], 1500)
    prm = scene_params(ws, num_threads=4, min_score_diff=-1e30, num_selected=10000)
    d, o = capi.Detector(**prm), api.Oracle(**prm)
    w = weights.make_lenet_weights(4)
    for x in (d, o):
        x.lenet_load(w)
        x.set_cloud(xyz)
        x.compute_normals()
    sel0, all0 = d.detect(sample_idx=idx, seed=2)
    for k in (1, 2):
        d.set_min_inliers(k)
        o.set_min_inliers(k)
        sel_d, all_d = d.detect(sample_idx=idx, seed=2)
        sel_o, all_o = o.detect(sample_idx=idx, seed=2)
        assert len(all_d) == len(all_o) == len(all0)
        assert 0 < len(sel_o) < len(sel0)
        # the clustering consumes LeNet scores, which agree to fp32 tolerance, not bit for bit:
        # same hands kept (geometric test only), positions bit-equal, mean scores within tolerance
        kd = sorted(zip(sel_d["sample_slot"].tolist(), sel_d["orientation"].tolist()))
        ko = sorted(zip(sel_o["sample_slot"].tolist(), sel_o["orientation"].tolist()))
        assert kd == ko
        od = np.lexsort((sel_d["orientation"], sel_d["sample_slot"]))
        oo = np.lexsort((sel_o["orientation"], sel_o["sample_slot"]))
        for f in ("bottom", "top", "surface", "axis"):
            assert np.array_equal(sel_d[f][od], sel_o[f][oo]), f
        tol = 1e-4 * np.abs(all_o["score"]).max() + 2e-3
        assert np.abs(sel_d["score"][od] - sel_o["score"][oo]).max() <= tol
        # and the C-ABI's standalone entry point on the detector's own scored list gives the same
        again = d.find_clusters(all_d, k)
        assert sorted(zip(again["sample_slot"].tolist(), again["orientation"].tolist())) == kd
    d.set_min_inliers(0)
    assert d.detect(sample_idx=idx, seed=2)[0].tobytes() == sel0.tobytes()
    d.close()
