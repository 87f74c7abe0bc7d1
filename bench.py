#!/usr/bin/env python3
"""bench.py -- headline benchmark of the agile_grasp2 hot path on MI355X.

Metric (BASELINE.json): grasp hypotheses scored per second on a 300k-point voxelised cloud
(config 2: num_samples=5000, 8 orientations, one GPU), end to end:
    search grid -> PCA normals -> local frames -> hand sweep -> prune -> grasp images -> LeNet ->
    score threshold / top-k
with the raw xyz cloud already resident in HBM when the timed region starts.  One "step" is one
full pass over one cloud.  value = hypotheses scored by all ranks / max-over-ranks wall time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg1|cfg3|cfg5] [--no-cpu]

--config cfg5 (BASELINE.json configuration 5): a 30 Hz stream -- `steps` frames of ONE scene whose
objects drift a few millimetres per frame, 2 000 samples per frame, through ag2_detect_frame (the
per-frame pipeline captured in a hipGraph); per-frame latency p50 / p99 against the 33.3 ms budget,
graph replay beside the step-by-step path.

N > 1 (launched by torch.distributed.run, one rank per GPU): the cloud is cut into spatial tiles.
The N x num_samples samples are ordered along the cloud's longest axis once, rank r owns a contiguous
range of them and holds only the points of that interval plus a halo of nn_radius_hands + normals_radius, binned
against the whole cloud's minimum (ag2_set_grid_origin) -- its hypotheses are exactly those of an
unsplit run (tests/test_tiles.py).  Every rank sweeps num_samples samples (weak scaling) and the
fixed-slot candidate tables are exchanged with one RCCL all-gather.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" for the dominant kernel (live HIP-event
durations on the launch stream; algorithmic bytes/flops per SURVEY.md section 8d with the measured
K1, K2, P, H of the run) and "cpu_baseline" (the CPU oracle -- a restatement of the reference
algorithm, not PCL/FLANN/Caffe -- on a bounded sample of the same workload, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (n_points, num_samples, orientations, voxelised, scene kind)
    "cfg1": (50_000, 500, 8, True, "tabletop"),
    "cfg2": (300_000, 5000, 8, True, "tabletop"),
    "cfg3": (1_000_000, 20000, 16, False, "tabletop"),
    "cfg5": (300_000, 2000, 8, True, "stream"),
}
FRAME_BUDGET_MS = 1000.0 / 30.0   # BASELINE.json configs[4]: 30 Hz
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
CONV_FLOP = 2.0 * (20 * 56 * 56 * 75 + 50 * 24 * 24 * 500)   # 38.208 MFLOP / image
FC_FLOP = 2.0 * (500 * 7200 + 2 * 500)                        # 7.202 MFLOP / image
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 (v_mfma_f32_32x32x16_bf16)
# k_lenet_conv_x3b: v_mfma_f32_32x32x16_bf16 issued per image = conv1 3 bands x 42 tiles x 5 k-blocks x 3
# terms (the bands overlap: 126 tiles for 98 tiles' worth of output) + conv2 18 tiles x 2 channel halves
# x 32 k-blocks x 6 terms; 32*32*16*2 flop each
CONV_X3_ISSUED_FLOP = (3 * 42 * 5 * 3 + 18 * 2 * 32 * 6) * 32768.0
# k_lenet_fc1_x3: per 32-image tile 16 column tiles x 450 k-blocks x 6 terms (ip2 and the padding of
# the batch to 128 images not counted)
FC_X3_ISSUED_FLOP = 16 * 450 * 6 * 32768.0 / 32


class no_gc:
    """Python's cyclic collector must not run inside a timed region: with torch imported one generation-2
    collection of this harness takes 35 - 60 ms -- fifty headline steps (0.68 ms each), or one frame of the
    cfg5 stream past its 33.3 ms budget (what the driver's round-3 record showed: attributed with gc.callbacks
    in round 4).  The library itself is C++ and has no collector; a C++ caller never sees this."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.collect()
        gc.disable()
        return self

    def __exit__(self, *a):
        import gc
        if self.was:
            gc.enable()
        return False


def launch_params(ws, R):
    """launch/file_detect_grasps.launch:16-48 values -- except min_score_diff: the launch files' 300 .. 800
    presume the trained network, which the reference does not ship (.MISSING_LARGE_BLOBS); with the seeded
    Xavier weights scores are O(10) around zero, so the threshold is 0 (about half of the scored
    hypotheses pass and reach the clustering / top-k / multi-GPU merge)."""
    from agile_grasp2_amd import scene
    return dict(finger_width=0.01, hand_outer_diameter=0.09, hand_depth=0.06, hand_height=0.02,
                init_bite=0.01, nn_radius_taubin=0.01, nn_radius_hands=0.1, num_orientations=R,
                filter_half_grasps=0, min_aperture=0.03, max_aperture=0.08, min_score_diff=0.0,
                num_selected=30, cam_origin=[scene.CAMERA, scene.CAMERA], workspace=list(ws))


def cpu_baseline(xyz, ws, idx, R, weights, budget_s=12.0, max_reps=400, threads=None):
    """Oracle (CPU restatement of the reference algorithm) on the SAME workload: whole steps (grid +
    normals + detect over all samples), repeated until about budget_s seconds of CPU work have been
    spent, so the sample is bounded in time, not in size.  Threads: the box's CPU share for one GPU
    (16), or fewer if the host has fewer cores."""
    from oracle import api
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail)) if threads is None else threads
    prm = launch_params(ws, R)
    o = api.Oracle(**dict(prm, num_threads=cores))
    o.lenet_load(weights)
    reps, scored_tot, t_tot = 0, 0, 0.0
    stage = {"grid": 0.0, "normals": 0.0, "frames": 0.0, "hands": 0.0, "images": 0.0, "lenet": 0.0}
    while reps < max_reps and (reps == 0 or t_tot < budget_s):
        t0 = time.perf_counter()
        o.set_cloud(xyz)
        t1 = time.perf_counter()
        o.compute_normals()
        _, scored = o.detect(sample_idx=idx, seed=1, do_prune=True)
        t_tot += time.perf_counter() - t0
        c = o.counters()
        stage["grid"] += t1 - t0
        for k in ("normals", "frames", "hands", "images", "lenet"):
            stage[k] += getattr(c, "t_" + k)
        scored_tot += len(scored)
        reps += 1
    return {
        "value": scored_tot / t_tot if t_tot > 0 else 0.0, "unit": "hypotheses/s", "cores": cores,
        "kind": "port",
        "sample": (f"oracle (CPU restatement: grid radius search instead of FLANN, own PCA normals / image "
                   f"scatter / fp32 LeNet instead of PCL / OpenCV / Caffe; -O3 -fopenmp -DNDEBUG, {cores} threads) "
                   f"on the same cloud, samples and weights: {reps} full steps in {t_tot:.2f} s, "
                   f"{scored_tot // max(1, reps)} hypotheses scored per step"),
        "ms_per_step": t_tot / max(1, reps) * 1e3,
        "stage_ms_per_step": {k: round(v / max(1, reps) * 1e3, 3) for k, v in stage.items()},
    }


def validate_against_unsplit(capi, device, prm, weights, xyz_full, ordered, seed, merged, n_total, scored_per_step):
    """N > 1 self-validation (untimed, rank 0, after the other ranks have left): the UNSPLIT cloud with all the
    ranks' samples (the ordered list the tiles were cut from) on this one GPU, against what the last timed step's
    exchange + merge produced.  hand_search.cpp:194-228 (samples are independent, results concatenate in sample
    order) and grasp_detector.cpp:228-252 (threshold, top-k) say what must hold:
      records_byte_equal   every merged record is a scored record of the unsplit run, every field but the score
                           byte for byte;
      scores_within_tol    LeNet scores within 2e-4 max|score| + 2e-3 (ip1's split-K follows the batch size: a
                           tile's scores may differ from the unsplit run's in the last bits);
      selection_ok         the merged top-k holds every record the unsplit scores demand and none they rule out
                           (agile_grasp2_amd/selection_check.py: only records within 2 tol of the threshold or of
                           the cut may differ), in descending score order;
      counts_ok            the ranks' scored hypotheses sum to the unsplit run's, the merged list's length lies
                           between the records certainly above and possibly above the threshold.
    A first hardware record of an N-GPU run therefore says by itself whether its result is the reference's."""
    import torch
    from agile_grasp2_amd.selection_check import check_selection
    du = capi.Detector(device=device, **prm)
    du.set_stream(torch.cuda.current_stream().cuda_stream)
    du.lenet_load(weights)
    du.set_cloud(xyz_full)
    du.compute_normals()
    _, uall = du.detect(sample_idx=ordered, seed=seed, do_prune=True)
    du.close()
    out = {"unsplit_scored": int(len(uall)), "ranks_scored": int(round(scored_per_step)), "merged_records": int(n_total),
           "merged_selected": int(len(merged))}
    tol = 2e-4 * float(np.abs(uall["score"]).max() if len(uall) else 0.0) + 2e-3
    thr, k = float(prm["min_score_diff"]), int(prm["num_selected"])
    key = {(int(h["sample_slot"]), int(h["orientation"])): i for i, h in enumerate(uall)}
    fields = [f for f in uall.dtype.names if f not in ("score", "full_antipodal", "reserved")]
    same, close = True, True
    for h in merged:
        i = key.get((int(h["sample_slot"]), int(h["orientation"])))
        if i is None:
            same = False
            continue
        same = same and all(np.array_equal(h[f], uall[i][f]) for f in fields)
        close = close and abs(float(h["score"]) - float(uall[i]["score"])) <= tol
    out["records_byte_equal"], out["scores_within_tol"] = bool(same), bool(close)
    try:
        chk = check_selection(merged, uall, thr, k, tol, max_uncertain=max(12, len(uall) // 100), tag="N > 1 merge")
        out["selection_ok"], out["selection"] = True, chk
    except AssertionError as e:
        out["selection_ok"], out["selection_error"] = False, str(e)[:300]
    lo, hi = int((uall["score"] >= thr + 2 * tol).sum()), int((uall["score"] >= thr - 2 * tol).sum())
    out["counts_ok"] = bool(out["ranks_scored"] == len(uall) and lo <= n_total <= hi)
    out["ok"] = bool(same and close and out["selection_ok"] and out["counts_ok"])
    out["tol"] = tol
    return out


def stream_legs(seed, n_frames, n_warm, S=None, only=None, host_leg=True):
    """BASELINE.json configuration 5: `n_frames` (+ n_warm untimed) RAW frames of one drifting tabletop scene
    (~765 000 points each, ~300 000 voxels of 3 mm) through ag2_detect_frame_raw -- workspace filter, voxel
    grid and uniform sub-sampling on the device inside the per-frame pipeline, which is captured in a
    hipGraph -- beside the same fixed-shape sequence without a graph, the step-by-step calls
    (ag2_preprocess_cloud_device + ag2_subsample_uniformly + ag2_compute_normals + ag2_detect on a second
    context; must return the same bytes) and, for the front end's share, frames that arrive already
    preprocessed (ag2_detect_frame).  Latency = host time of one call: raw cloud in HBM -> selected grasps
    in host memory."""
    import torch
    from agile_grasp2_amd import capi, scene
    from agile_grasp2_amd.weights import make_lenet_weights
    n_points, S0, R, _, _ = CONFIGS["cfg5"]
    S = int(S or S0)
    raws, ws = scene.make_stream(seed, int(2.55 * n_points), n_frames + n_warm, voxel=None)
    prm = launch_params(ws, R)
    weights = make_lenet_weights(7)
    dev = [torch.from_numpy(c).cuda() for c in raws]
    torch.cuda.synchronize()

    def make():
        d = capi.Detector(**prm)      # own non-blocking stream (the legacy default stream cannot be captured)
        d.lenet_load(weights)
        d.set_stage_timing(0)
        return d

    # Attribution of a slow frame (one 58 ms frame in the driver's record, cause unknown): every
    # frame's latency is kept, with the library's own clock around its submit half (pack kernel + graph launch)
    # and its wait half (polling the flag behind the results), the number of waits that fell back from polling
    # to the stream, and the time Python's cyclic collector ran inside the frame (gc.callbacks) -- the harness
    # is a Python process, the library is not.
    import gc
    gc_state = {"t0": 0.0, "ms": 0.0, "runs": 0, "gen2": 0}

    def gc_cb(phase, info):
        if phase == "start":
            gc_state["t0"] = time.perf_counter()
        else:
            gc_state["ms"] += (time.perf_counter() - gc_state["t0"]) * 1e3
            gc_state["runs"] += 1
            gc_state["gen2"] += 1 if info.get("generation") == 2 else 0

    gc.callbacks.append(gc_cb)
    gc_off = os.environ.get("AG2_BENCH_GC", "off") != "on"   # default: no collection inside a latency leg
    legs, results, pre = {}, {}, {}
    for name in ("graph", "plain", "stepwise", "graph_preprocessed"):
        if only and name != only:
            continue
        if name == "graph_preprocessed" and "stepwise" not in results:
            continue
        d = make()
        if name != "stepwise":
            d.stream_configure(0, 0, name != "plain")
        lat, scored, out, vox = [], 0, [], 0
        sub_us, wait_us, gc_ms = [], [], []
        if gc_off:
            gc.collect()
            gc.disable()
        for k in range(n_frames + n_warm):
            if name == "graph_preprocessed":   # the frames as the stepwise leg's front end left them, resident in HBM
                cloud, idx = pre[k]
            gc0 = gc_state["ms"]
            t0 = time.perf_counter()
            if name == "stepwise":
                vox = d.preprocess_cloud_device(dev[k].data_ptr(), raws[k].shape[0], 12, voxel_size=scene.VOXEL)
                ns = d.subsample_uniformly(S, seed=seed + k, want_indices=False)
                d.compute_normals()
                sel, n_sc = d.detect(n_resident=ns, seed=seed, do_prune=True, want_all=False)
            elif name == "graph_preprocessed":
                sel, n_sc = d.detect_frame(sample_idx=idx, seed=seed, do_prune=True, dptr=cloud.data_ptr(),
                                           n=cloud.shape[0], stride=12)
            else:
                sel, n_sc, vox = d.detect_frame_raw(num_samples=S, sample_seed=seed + k, seed=seed, do_prune=True,
                                                    dptr=dev[k].data_ptr(), n=raws[k].shape[0], stride=12,
                                                    voxel_size=scene.VOXEL)
            dt = time.perf_counter() - t0
            out.append(sel.tobytes())
            if name == "stepwise" and not only:   # (kept on the host until the leg is over: a device allocation
                pre[k] = (d.get_cloud()[0], d.get_samples())   #  between two frames can stall the next one)
            if k >= n_warm:
                lat.append(dt * 1e3)
                scored += n_sc
                gc_ms.append(gc_state["ms"] - gc0)
                if name != "stepwise":
                    wi = d.wait_info()
                    sub_us.append(int(wi.last_submit_us))
                    wait_us.append(int(wi.last_wait_us))
        if gc_off:
            gc.enable()
        lat = np.array(lat)
        legs[name] = {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
                      "max_ms": float(lat.max()), "max_frame": int(lat.argmax()), "mean_ms": float(lat.mean()),
                      "scored_per_s": scored / (lat.sum() * 1e-3), "scored_per_frame": scored / n_frames,
                      "voxels_last_frame": int(vox), "lat_ms": [round(float(v), 3) for v in lat],
                      "gc_ms_in_frames": round(float(sum(gc_ms)), 3), "gc_disabled_in_leg": bool(gc_off)}
        if name != "stepwise":
            fi = d.frame_info()
            legs[name]["frame_info"] = {f: int(getattr(fi, f)) for f, _ in fi._fields_}
            wi = d.wait_info()
            legs[name]["host_split"] = {"submit_us": sub_us, "wait_us": wait_us,
                                        "poll_fallbacks": int(wi.poll_fallbacks), "poll_yields": int(wi.poll_yields),
                                        "note": ("library clock: submit = pack kernel + hipGraphLaunch (or the kernel-by-"
                                                 "kernel sequence), wait = polling the flag k_topk writes behind the "
                                                 "results; lat_ms - (submit + wait) = the Python harness")}
        results[name] = out
        legs[name]["hypotheses_last_frame"] = int(d.counters().n_hypotheses)
        d.close()
        if name == "stepwise" and not only:
            pre = {k: (torch.from_numpy(c).cuda(), i) for k, (c, i) in pre.items()}
            torch.cuda.synchronize()
    gc.callbacks.remove(gc_cb)
    if only:
        return {"leg": only, **legs[only]}
    same = (results["graph"] == results["stepwise"] and results["plain"] == results["stepwise"]
            and results["graph_preprocessed"] == results["stepwise"])
    host = None
    if host_leg:   # frames handed over in HOST memory (the PCIe-inclusive figure, never `value`)
        d = make()
        d.stream_configure(0, 0, True)
        lat = []
        for k in range(n_frames + n_warm):
            t0 = time.perf_counter()
            d.detect_frame_raw(raws[k], num_samples=S, sample_seed=seed + k, seed=seed, do_prune=True,
                               voxel_size=scene.VOXEL)
            if k >= n_warm:
                lat.append((time.perf_counter() - t0) * 1e3)
        d.close()
        host = {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
                "note": f"raw frames handed over in pageable host memory ({raws[0].nbytes / 1e6:.1f} MB H2D per frame)"}
    g = legs["graph"]
    return {"budget_ms": FRAME_BUDGET_MS, "graph": g, "plain_fixed_shape": legs["plain"], "stepwise": legs["stepwise"],
            "graph_preprocessed_frames": legs["graph_preprocessed"], "graph_host_frames": host,
            "front_end_ms_p50": g["p50_ms"] - legs["graph_preprocessed"]["p50_ms"],
            "within_budget": bool(g["max_ms"] <= FRAME_BUDGET_MS), "same_bytes_as_stepwise": bool(same),
            "gc": {"collections_during_legs": gc_state["runs"], "gen2": gc_state["gen2"], "total_ms": round(gc_state["ms"], 3)},
            "raw_points_per_frame": [int(c.shape[0]) for c in raws[n_warm:n_warm + 4]] + ["..."],
            "num_samples_per_frame": S, "num_orientations": R, "frames": n_frames, "warmup_frames": n_warm}


def bench_stream(args):
    """--config cfg5: the frame stream as its own bench line (see stream_legs)."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    n_points, S, R, _, _ = CONFIGS["cfg5"]
    S = int(os.environ.get("AG2_STREAM_S", S))   # experiment aid: other sample counts on the same stream
    n_frames, n_warm = args.steps, max(3, args.warmup)   # warm-up: step-by-step, plain run + capture, first replay
    only = os.environ.get("AG2_STREAM_LEG")   # profiling aid: run one leg only
    lat = stream_legs(args.seed, n_frames, n_warm, S=S, only=only)
    if only:
        print(json.dumps(lat), flush=True)
        return
    g = lat["graph"]
    out = {
        "metric": "grasp hypotheses scored/sec on 300k-pt cloud; end-to-end detect latency",
        "value": g["scored_per_s"], "unit": "hypotheses/s", "n_gpus": 1, "steps": n_frames, "warmup": n_warm,
        "ms_per_step": g["mean_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 geometry + f32 LeNet (3 x bf16 operand split on bf16 MFMA, fp32 accumulate)",
        "data": "synthetic", "span": "hbm-resident (RAW frames: workspace filter + voxel grid + sub-sampling inside the frame)",
        "config": {"workload": (f"cfg5: stream of {n_frames} RAW frames (~{int(2.55 * n_points)} points, ~{n_points} voxels of 3 mm) of "
                                f"a tabletop scene whose objects drift 4 mm per frame, num_samples={S} per frame, {R} "
                                f"orientations, launch-file hand geometry, seeded LeNet weights; LeNet list padded to "
                                f"whole batches of 256 images; filter + voxel grid + sub-sampling + detect captured in "
                                f"one hipGraph (ag2_detect_frame_raw)"),
                   "raw_points_per_frame": lat["raw_points_per_frame"],
                   "num_samples_per_frame": S, "num_orientations": R, "batch_size": 256},
        "latency": lat,
        "roofline": None, "cpu_baseline": None,
        "note": "roofline / cpu_baseline are reported on the headline configuration (python bench.py); this line is the streaming latency",
    }
    print(json.dumps(out), flush=True)


def side_cfg3(args, weights, local_rank, steps=5):
    """BASELINE.json configuration 3 (1 M un-voxelised points, 20 000 samples, 16 orientations) as a side leg of
    the default line: `steps` timed steps, the sweep's roofline from its HIP events and the committed counter
    passes (profiles/pmc_traffic.json)."""
    import torch
    from agile_grasp2_amd import capi, scene
    n_points, S, R, voxelised, kind = CONFIGS["cfg3"]
    xyz, ws = scene.make_scene(args.seed, n_points, kind=kind, voxel=None)
    idx = scene.draw_samples(args.seed, xyz.shape[0], S)
    d = capi.Detector(device=local_rank, **launch_params(ws, R))
    d.set_stream(torch.cuda.current_stream().cuda_stream)
    d.lenet_load(weights)
    d.set_stage_timing(1)
    xyz_dev = torch.from_numpy(xyz).cuda()
    torch.cuda.synchronize()

    def step():
        d.set_cloud_device(xyz_dev.data_ptr(), xyz.shape[0], 12)
        d.compute_normals()
        return d.detect(sample_idx=idx, seed=args.seed, do_prune=True, want_all=False)[1]

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with no_gc():
        t0, scored, sweep_ms = time.perf_counter(), 0, 0.0
        for _ in range(steps):
            scored += step()
            t = d.times()
            sweep_ms += t.sweep_ms + t.sweep_overflow_ms
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    sweep_ms /= steps
    c = d.counters()
    d.close()
    dd = capi.Detector(device=local_rank, **dict(launch_params(ws, R), debug_flags=1))   # exact K2 (no row culling)
    dd.set_stream(torch.cuda.current_stream().cuda_stream)
    dd.lenet_load(weights)
    dd.set_cloud_device(xyz_dev.data_ptr(), xyz.shape[0], 12)
    dd.compute_normals()
    dd.detect(sample_idx=idx, seed=args.seed, do_prune=True, want_all=False)
    sum_k2 = dd.counters().sum_k2
    dd.close()
    work = sum_k2 * 24 + c.n_hypotheses * 176 + c.sum_p * 24
    achieved = work / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
    traffic, src = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        pk = pmc.get("cfg3", {})
        if "k_sweep" in pk:
            traffic = pk.get("k_sweep", 0) + pk.get("k_sweep_overflow", 0) + pk.get("k_sweep_orient", 0)
            src = f"profiles/{pmc.get('_profile', {}).get('cfg3', '?')} (committed rocprofv3 counter passes, not this run)"
    except Exception:
        pass
    return {"value": scored / steps / dt, "unit": "hypotheses/s", "ms_per_step": dt * 1e3, "steps": steps,
            "workload": f"cfg3: {xyz.shape[0]}-pt un-voxelised tabletop cloud, num_samples={S}, {R} orientations",
            "roofline": {"kernel": "k_sweep", "bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": achieved / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": src,
                         "launch_ms": sweep_ms, "algorithmic_work_per_launch": work},
            "counters": {"hypotheses": int(c.n_hypotheses), "scored": int(c.n_scored), "overflow_samples": int(c.n_overflow_samples),
                         "mean_K2": sum_k2 / max(1, c.n_frames), "mean_Kcrop": c.sum_kcrop / max(1, c.n_frames),
                         "mean_P": c.sum_p / max(1, c.n_hypotheses), "list_points": int(c.list_points)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--force-dist", action="store_true",
                    help="run the RCCL exchange path even with one rank (rehearsal on a 1-GPU box)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="N > 1: weak = num_samples per GPU fixed (default, what the driver's scaling run uses), "
                         "strong = the configuration's num_samples in total, split over the GPUs")
    args = ap.parse_args()
    if args.config == "cfg5":
        if args.steps == 10:
            args.steps = 30   # 30 frames
        return bench_stream(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # started by hand without torch.distributed.run: start it as a child process
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", "29517",
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 control flow on a box with ONE GPU (RCCL refuses two ranks on one
    # device): AG2_BENCH_REHEARSAL=1 puts every rank on GPU 0 and runs the collectives over gloo with
    # the buffers staged through host memory.  Timings of such a run mean nothing.
    rehearsal = os.environ.get("AG2_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    # Exactly ONE line goes to stdout (the JSON, rank 0).  Libraries print banners there (RCCL's
    # version block, for one), so fd 1 points at stderr until the result is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch  # plumbing only: device memory, stream, torch.distributed (RCCL)
    import torch.distributed as dist
    from agile_grasp2_amd import capi, scene
    from agile_grasp2_amd.weights import make_lenet_weights

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29518")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    coll_dev = "cpu" if rehearsal else "cuda"   # where tensors handed to collectives live

    n_points, S, R, voxelised, kind = CONFIGS[args.config]
    xyz, ws = scene.make_scene(args.seed, n_points, kind=kind, voxel=scene.VOXEL if voxelised else None)
    from agile_grasp2_amd import sharding
    prm = launch_params(ws, R)
    n_cloud = xyz.shape[0]
    xyz_full = xyz
    origin, slot_base, tile_note, ordered = None, 0, None, None
    if dist_on:
        # spatial tiles: same scene and same N x S samples on every rank, cut by x
        axis = sharding.longest_axis(xyz)
        s_total = S * world if args.scaling == "weak" else S
        ordered = sharding.order_samples_by_x(xyz, scene.draw_samples(args.seed, n_cloud, s_total), axis)
        halo = sharding.tile_halo(prm["nn_radius_hands"], prm["nn_radius_taubin"], 0.01)
        # tiles balanced on the samples' neighbour counts, not on their number
        bounds = sharding.balanced_bounds(sharding.sample_costs(xyz, ordered, prm["nn_radius_hands"], axis), world)
        keep, idx, slot_base = sharding.tile_points(xyz, ordered, rank, world, halo, axis, bounds)
        origin = sharding.cloud_origin(xyz)
        xyz = np.ascontiguousarray(xyz[keep])
        tile_note = f"{xyz.shape[0]} of {n_cloud} points on rank 0"
    else:
        idx = scene.draw_samples(args.seed, n_cloud, S)
    weights = make_lenet_weights(7)
    d = capi.Detector(device=local_rank, **prm)
    assert abs(d.params.normals_radius - 0.01) < 1e-12     # the halo above assumes the default
    d.set_stream(torch.cuda.current_stream().cuda_stream)
    d.lenet_load(weights)
    d.set_grid_origin(origin)
    xyz_dev = torch.from_numpy(xyz).cuda()          # HBM-resident input (this rank's tile when N > 1)
    torch.cuda.synchronize()
    # exchange buffer: the candidates in compact form (header + occupied slots, sharding.py); its
    # capacity is fixed after the warm-up from the largest list any rank produced
    s_rank = len(idx)                                # this rank's samples
    # exchange capacity: the same on every rank (the largest tile's slots) -- all-gather needs equal sizes
    s_max_rank = int(np.diff(bounds).max()) if dist_on else s_rank
    xch = {"cap": max(1, s_max_rank * R), "buf": None}

    def size_exchange(cap):
        xch["cap"] = int(cap)
        xch["buf"] = torch.empty(sharding.compact_bytes(xch["cap"]), dtype=torch.uint8, device="cuda")

    if dist_on:
        size_exchange(xch["cap"])

    acc = {}

    def local_step(local_select=True):  # this rank's part of a step: no collective
        d.set_cloud_device(xyz_dev.data_ptr(), xyz.shape[0], 12)
        d.compute_normals()
        # (a rank of a multi-GPU run leaves the top-k to the merge: no local read-back)
        sel, n_scored = d.detect(sample_idx=idx, slot_base=slot_base, seed=args.seed, do_prune=True,
                                 want_all=False, local_select=local_select)
        return n_scored

    import ctypes
    cnt_rec = capi.Counters()
    get_cnt, cnt_ref = d.L.ag2_get_counters, ctypes.byref(cnt_rec)

    def step():
        if not dist_on:
            return local_step(local_select=True)
        # A rank's detect waits for nothing (from its second call on: the tail is launched at the shapes the
        # previous call left, and whether they held travels in the exported header).  Then the path's one exchange
        # step: every rank's scored candidates above the threshold (compact form), RCCL all-gather over xGMI; every
        # rank merges -- the lists concatenated in rank (= sample) order, top num_selected by score
        # (grasp_detector.cpp:239-252) -- on the device.  AG2_ERR_RETRY (every rank reads the same headers, so every
        # rank gets it): some rank's shapes did not hold; all repeat the step, inside the timed region.
        for attempt in range(3):
            local_step(local_select=False)
            d.export_selected_compact_device(xch["buf"].data_ptr(), xch["buf"].numel(), xch["cap"])
            if rehearsal:
                xch["out"] = sharding.all_gather_tables(xch["buf"].cpu(), world).cuda()
            else:
                xch["out"] = sharding.all_gather_tables(xch["buf"], world, out=xch.get("out"))
            try:
                xch["merged"], xch["n_total"] = d.merge_selected_device(xch["out"].data_ptr(), world, xch["cap"])
            except capi.RetryStep:
                xch["retries"] = xch.get("retries", 0) + 1
                continue
            get_cnt(d.h, cnt_ref)   # (the count is known once the merge has waited for the stream; bare C call)
            return int(cnt_rec.n_scored)
        raise SystemExit("a rank's shapes did not settle in three attempts")

    def sync():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # Timed region: only the HIP events around the sweep (the dominant kernel) are recorded -- every
    # stage event costs a few microseconds of stream serialisation, all of them about 3 % of a step.
    # The other stages' durations come from an extra, untimed pass below.
    d.set_stage_timing(1)
    for _ in range(max(1, args.warmup) if dist_on else args.warmup):
        step()
    if dist_on:
        # every rank sizes the exchange for twice the largest selected list seen (same on all ranks)
        hdr0 = xch["buf"][:4].cpu().numpy().view(np.uint32)[0]
        nh = torch.tensor([float(hdr0)], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(nh, op=dist.ReduceOp.MAX)
        size_exchange(min(s_max_rank * R, max(1024, 2 * int(nh.item()))))
        step()
    sync()
    acc_sweep = [0.0, 0.0]
    # (only the sweep's events are recorded in here; the stage times are read into ONE preallocated record
    # through the bare C call: building a ctypes structure per step was ~5 us of idle GPU between steps)
    import ctypes
    t = capi.Times()
    read_times, t_ref = d.L.ag2_get_stage_times, ctypes.byref(t)
    with no_gc():   # (the collector of this Python harness is not part of the step)
        sync()
        t0 = time.perf_counter()
        scored = 0
        for _ in range(args.steps):
            scored += step()
            read_times(d.h, t_ref)
            acc_sweep[0] += t.sweep_ms
            acc_sweep[1] += t.sweep_overflow_ms
        sync()
        elapsed = time.perf_counter() - t0
    acc["sweep_ms"], acc["sweep_overflow_ms"] = acc_sweep
    c = d.counters()

    if dist_on:  # no rank's list may have been cut (the headers say): else the bench line is void
        hdr = xch["out"].view(world, -1)[:, :8].cpu().numpy().view(np.uint32)
        if (hdr[:, 0] > hdr[:, 1]).any():
            raise SystemExit(f"exchange capacity {xch['cap']} too small: {hdr[:, 0].tolist()}")
    sweep_rank_ms = (acc.get("sweep_ms", 0.0) + acc.get("sweep_overflow_ms", 0.0)) / max(1, args.steps)
    tt = torch.tensor([elapsed, float(scored), float(c.n_hypotheses), float(xyz.shape[0]), float(s_rank),
                       sweep_rank_ms], dtype=torch.float64, device=coll_dev)
    per_rank = None
    if dist_on:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tmin = tt.clone()
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        per_rank = {"points_max": int(tmax[3]), "points_min": int(tmin[3]), "points_sum": int(tt[3]),
                    "halo_duplication": float(tt[3]) / n_cloud,   # sum of the tiles' points / cloud points
                    "samples_max": int(tmax[4]), "samples_min": int(tmin[4]),
                    "sweep_ms_max": float(tmax[5]), "sweep_ms_mean": float(tt[5]) / world,
                    "merged_records": int(xch.get("n_total", 0)), "merged_selected": int(len(xch.get("merged", []))),
                    "steps_repeated_rank0": int(xch.get("retries", 0)),
                    "rank0_detect_one_trip": int(c.detect_one_trip), "rank0_detect_redone": int(c.detect_redone)}
    total_scored = float(tt[1])
    if rank != 0:
        dist.destroy_process_group()
        return
    if dist_on:
        per_rank["matches_unsplit"] = validate_against_unsplit(capi, local_rank, prm, weights, xyz_full, ordered, args.seed,
                                                              xch["merged"], int(xch["n_total"]), total_scored / args.steps)

    # Untimed diagnostic pass: the measured path culls stencil rows by the sphere and the crop slab
    # and so never visits all K2 radius neighbours; the roofline's algorithmic bytes are defined on
    # K2, which debug_flags=1 counts exactly.
    dd = capi.Detector(device=local_rank, **dict(launch_params(ws, R), debug_flags=1))
    dd.set_stream(torch.cuda.current_stream().cuda_stream)
    dd.lenet_load(weights)
    dd.set_grid_origin(origin)
    dd.set_cloud_device(xyz_dev.data_ptr(), xyz.shape[0], 12)
    dd.compute_normals()
    dd.detect(sample_idx=idx, slot_base=slot_base, seed=args.seed, do_prune=True, want_all=False)
    sum_k2 = dd.counters().sum_k2
    dd.close()

    K = args.steps
    ms = {k: v / K for k, v in acc.items() if not k.startswith("reserved")}
    # untimed pass with every stage event on: durations of the stages other than the sweep
    d.set_stage_timing(2)
    K2, acc2 = max(3, min(K, 10)), {}
    for _ in range(K2):
        local_step()  # (rank 0 alone by now: no collective here)
        t = d.times()
        for name, _ty in t._fields_:
            acc2[name] = acc2.get(name, 0.0) + getattr(t, name)
    torch.cuda.synchronize()
    for k_, v_ in acc2.items():
        if k_ not in ("sweep_ms", "sweep_overflow_ms") and not k_.startswith("reserved"):
            ms[k_] = v_ / K2
    # the timed region records the sweep as a whole (two events); how it divides into the part up to the
    # gates and the rest (second stage, orientation kernel) is taken from the untimed pass
    sweep_total = ms["sweep_ms"] + ms["sweep_overflow_ms"]
    a2, b2 = acc2.get("sweep_ms", 0.0), acc2.get("sweep_overflow_ms", 0.0)
    if a2 + b2 > 0:
        ms["sweep_ms"], ms["sweep_overflow_ms"] = sweep_total * a2 / (a2 + b2), sweep_total * b2 / (a2 + b2)
    n_img = c.n_scored
    # algorithmic work per launch, measured neighbourhood sizes of this run
    kernels = {
        "k_normals": dict(bound="hbm", work=c.sum_k1 * 12 + c.n_valid_points * 12, ms=ms["normals_ms"]),
        "k_sweep": dict(bound="hbm", work=sum_k2 * 24 + c.n_hypotheses * 176 + c.sum_p * 24,
                        ms=ms["sweep_ms"] + ms["sweep_overflow_ms"]),  # both instantiations
        "k_render": dict(bound="hbm", work=c.sum_p * 24 * (n_img / max(1, c.n_hypotheses)) + n_img * 10800,
                         ms=ms["render_ms"]),
        # the convolutions run on the bf16 matrix cores with every fp32 operand split into three exact
        # bf16 terms (k_lenet_x3.hip): priced by the bf16 MFMA work actually ISSUED (3 terms for conv1,
        # 6 for conv2, padding included) against the dense bf16 peak; AG2_LENET_F32=1 selects the
        # f32-input MFMA kernel, priced by the useful fp32 flops against the fp32 MFMA peak
        "k_lenet_conv": (dict(bound="mfma", work=n_img * CONV_FLOP, ms=ms["lenet_conv_ms"])
                         if os.environ.get("AG2_LENET_F32") else
                         dict(bound="mfma", work=n_img * CONV_X3_ISSUED_FLOP, ms=ms["lenet_conv_ms"],
                              peak=PEAK_BF16_MFMA_TFLOPS,
                              fp32_equivalent_tflops=n_img * CONV_FLOP / (max(ms["lenet_conv_ms"], 1e-9) * 1e-3) / 1e12)),
        "k_lenet_fc": (dict(bound="mfma", work=n_img * FC_FLOP, ms=ms["lenet_fc_ms"])
                       if os.environ.get("AG2_LENET_F32") else
                       dict(bound="mfma", work=n_img * FC_X3_ISSUED_FLOP, ms=ms["lenet_fc_ms"],
                            peak=PEAK_BF16_MFMA_TFLOPS,
                            fp32_equivalent_tflops=n_img * FC_FLOP / (max(ms["lenet_fc_ms"], 1e-9) * 1e-3) / 1e12)),
    }
    for k in kernels.values():
        if k["bound"] == "hbm":
            k["achieved"] = k["work"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
            k["peak"], k["unit"] = PEAK_HBM_GBS, "GB/s"
        else:
            k["achieved"] = k["work"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
            k["peak"], k["unit"] = k.get("peak", PEAK_F32_MFMA_TFLOPS), "TFLOP/s"
        k["frac"] = k["achieved"] / k["peak"]
        if "fp32_equivalent_tflops" in k:
            # The three-term split issues about 6 bf16 multiply-adds (padding included: 7.5) per useful fp32
            # one.  `frac` counts USEFUL flops only (fp32-equivalent rate / dense bf16 peak of the pipes the
            # kernel runs on); the pipes' own utilisation is issued_frac; useful_vs_f32_mfma_peak compares with
            # what the f32-input matrix instructions could do at best.
            k["issued_tflops"], k["issued_frac"] = k["achieved"], k["frac"]
            k["achieved"] = k["fp32_equivalent_tflops"]
            k["frac"] = k["achieved"] / k["peak"]
            k["useful_vs_f32_mfma_peak"] = k["achieved"] / PEAK_F32_MFMA_TFLOPS
    dom = max(kernels, key=lambda n: kernels[n]["ms"])
    # HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes of the same command
    # (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024
    # with the gfx950 factor-2 correction of MI355X_MICROARCH.md).  Counters cannot be read from
    # inside this process, so the committed summary of the last profiled run is quoted; null if the
    # kernel is not in it.
    traffic, traffic_source, write_amp = None, None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
            # MFMA-busy fraction of the matrix-core kernels (SQ_VALU_MFMA_BUSY_CYCLES over the busy
            # CUs' SIMD cycles) from the committed counter pass of the same command
            if args.config == "cfg2" and not os.environ.get("AG2_LENET_F32"):
                for name, frac in pmc.get("mfma_busy", {}).items():
                    if name in kernels:
                        kernels[name]["mfma_busy"] = frac
            per_kernel = pmc.get(args.config, {})
            traffic = per_kernel.get(dom)
            if dom == "k_sweep" and traffic is not None:  # all of the sweep's kernels, like the duration
                traffic += per_kernel.get("k_sweep_overflow", 0) + per_kernel.get("k_sweep_orient", 0)
            if traffic is not None:
                traffic_source = (f"profiles/{pmc.get('_profile', {}).get(args.config, '?')} via profiles/pmc_traffic.json: "
                                  "committed rocprofv3 counter passes of this command (FETCH_SIZE and WRITE_SIZE "
                                  "in separate runs), NOT measured by this run")
            # bytes the dominant kernel HAS to write (records + point lists for the sweep) against the
            # WRITE_SIZE counter: register spills and scratch traffic show up as the excess
            wr = pmc.get("_writes", {}).get(args.config, {})
            if "k_sweep" in wr:   # records + point lists + (split sweep) the cropped lists of the passing samples
                must = c.n_hypotheses * 176 + c.sum_p * 48 + c.n_hypotheses * 9 + c.list_points * 16
                sweep_wa = (wr["k_sweep"] + wr.get("k_sweep_overflow", 0) + wr.get("k_sweep_orient", 0)) / max(1, must)
                kernels["k_sweep"]["write_amplification"] = round(sweep_wa, 3)
                kernels["k_sweep"]["traffic"] = (per_kernel.get("k_sweep", 0) + per_kernel.get("k_sweep_overflow", 0) +
                                                 per_kernel.get("k_sweep_orient", 0))
                if dom == "k_sweep":
                    write_amp = sweep_wa
        except Exception:
            traffic = None
    roofline = {"kernel": dom, "bound": kernels[dom]["bound"], "achieved": kernels[dom]["achieved"],
                "peak": kernels[dom]["peak"], "unit": kernels[dom]["unit"], "frac": kernels[dom]["frac"],
                "traffic": traffic, "traffic_source": traffic_source, "write_amplification": write_amp,
                "launch_ms": kernels[dom]["ms"],
                "algorithmic_work_per_launch": kernels[dom]["work"],
                "all_kernels": {n: dict({"ms": round(k["ms"], 4), "achieved": round(k["achieved"], 3),
                                         "unit": k["unit"], "peak": k["peak"], "frac": round(k["frac"], 4)},
                                        **({"issued_tflops": round(k["issued_tflops"], 2),
                                            "issued_frac": round(k["issued_frac"], 4),
                                            "useful_vs_f32_mfma_peak": round(k["useful_vs_f32_mfma_peak"], 4)}
                                           if "fp32_equivalent_tflops" in k else {}),
                                        **({"mfma_busy": k["mfma_busy"]} if "mfma_busy" in k else {}),
                                        **({"write_amplification": k["write_amplification"], "traffic": k["traffic"]}
                                           if "write_amplification" in k else {}))
                                for n, k in kernels.items()}}
    out = {
        "metric": "grasp hypotheses scored/sec on 300k-pt cloud; end-to-end detect latency",
        "value": total_scored / elapsed, "unit": "hypotheses/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": ("f64 geometry + f32 LeNet (fp32 MFMA)" if os.environ.get("AG2_LENET_F32") else
                  "f64 geometry + f32 LeNet (conv1, conv2, ip1: every fp32 operand as the exact sum of 3 bf16 "
                  "terms on bf16 MFMA, fp32 accumulate; ip2 in fp32)"),
        "data": "synthetic",
        "span": ("hbm-resident: the cloud is in HBM when the timed region starts, the selected grasps are in host memory "
                 "when it ends (pcie_inclusive.value: the same step with the cloud handed over in host memory)"),
        "config": {
            "workload": (f"{args.config}: {n_cloud}-pt {'voxelised (3 mm)' if voxelised else 'un-voxelised'} "
                         f"synthetic tabletop cloud, num_samples={S}"
                         f"{'/GPU' if (args.scaling == 'weak' or not dist_on) else ' in total, shared by the GPUs'}, "
                         f"{R} orientations, launch-file hand "
                         f"geometry, seeded LeNet weights"),
            "n_points": int(n_cloud), "num_samples_per_gpu": (S if args.scaling == "weak" or not dist_on else S // world),
            "num_orientations": R, "per_rank": per_rank,
            "hypotheses_per_step_per_gpu": int(c.n_hypotheses), "scored_per_step_per_gpu": int(n_img),
            "slots_swept_per_s": S * R * (world if (args.scaling == "weak" or not dist_on) else 1) / (elapsed / K),
            "mean_K1": c.sum_k1 / max(1, c.n_valid_points), "mean_K2": sum_k2 / max(1, c.n_frames),
            "mean_Kcrop": c.sum_kcrop / max(1, c.n_frames), "mean_P": c.sum_p / max(1, c.n_hypotheses),
            "overflow_samples": int(c.n_overflow_samples),
            "detect_one_trip": int(c.detect_one_trip), "detect_redone": int(c.detect_redone),
            "parallelism": ("single GPU" if not dist_on else
                            f"{world} spatial tiles along the cloud's longest axis, cut so that the summed "
                            f"neighbour counts of their samples are equal (interval of the rank's samples + 0.11 m "
                            f"halo, {tile_note}); one RCCL all-gather of every rank's scored candidates above the "
                            f"threshold in compact form (16 B header + up to {xch['cap']} records x 176 B per "
                            f"rank), then the global top-{prm['num_selected']} by score on every rank, inside the "
                            f"timed step"),
        },
        "stage_ms": {k: round(v, 4) for k, v in ms.items()},
        "stage_ms_note": ("sweep_ms and sweep_overflow_ms: HIP events inside the timed region; the other stages: "
                          f"an extra untimed pass of {K2} steps with all stage events on (they cost ~3 % of a step)"),
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu:
        # Untimed side legs (never `value`):
        # (1) the same step with the cloud handed over in HOST memory (PCIe-inclusive rate);
        def timed(fn, reps=10):
            fn()
            torch.cuda.synchronize()
            with no_gc():
                t1 = time.perf_counter()
                tot = 0
                for _ in range(reps):
                    tot += fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / reps, tot / reps

        d.set_stage_timing(1)  # the side legs are throughput figures like the timed region

        def host_step():
            d.set_cloud(xyz)
            d.compute_normals()
            return d.detect(sample_idx=idx, seed=args.seed, do_prune=True, want_all=False)[1]

        dt, sc = timed(host_step)
        out["pcie_inclusive"] = {"value": sc / dt, "unit": "hypotheses/s", "ms_per_step": dt * 1e3,
                                 "note": f"cloud ({xyz.nbytes / 1e6:.1f} MB) uploaded from pageable host memory every step"}
        # (2) the step with the GPU front end in front of it: RAW (un-voxelised) cloud resident in HBM ->
        # workspace filter + 3 mm voxel grid + uniform sub-sampling -> normals -> detect, the processed
        # cloud and the sample indices never leaving the device -- through
        # ag2_detect_frame_raw (everything in one captured sequence, one host synchronisation), and through
        # the separate calls for comparison.
        if voxelised:
            # The raw cloud: 2 - 3 points inside every voxel of the headline cloud (those inside the workspace: the
            # front end filters), shuffled -- its voxelisation is the headline cloud again, so this leg differs
            # from the headline step by the front end (and by which samples the device draws).
            xd = xyz.astype(np.float64)
            inside = ((xd[:, 0] > ws[0]) & (xd[:, 0] < ws[1]) & (xd[:, 1] > ws[2]) & (xd[:, 1] < ws[3]) &
                      (xd[:, 2] > ws[4]) & (xd[:, 2] < ws[5]))
            raw, ws_raw = scene.raw_from_voxels(xyz[inside], args.seed), ws
            df = capi.Detector(**launch_params(ws_raw, R))   # own stream: the sequence is captured in a hipGraph
            df.lenet_load(weights)
            df.set_stage_timing(0)
            raw_dev = torch.from_numpy(raw).cuda()
            torch.cuda.synchronize()
            fe = {}

            def raw_frame_step():
                _, n_sc, fe["n_vox"] = df.detect_frame_raw(num_samples=S, sample_seed=args.seed, seed=args.seed,
                                                           do_prune=True, dptr=raw_dev.data_ptr(), n=raw.shape[0],
                                                           stride=12, voxel_size=scene.VOXEL)
                return n_sc

            for _ in range(3):   # step by step, fixed shapes + capture, first replay
                raw_frame_step()
            dt, sc = timed(raw_frame_step)
            fi = df.frame_info()
            # the same workload without the front end: the processed cloud and the drawn samples handed to the
            # plain step (set_cloud_device + normals + detect) of the headline measurement
            cloud_dev = torch.from_numpy(df.get_cloud()[0]).cuda()
            idx_fe = df.get_samples()
            dp = capi.Detector(device=local_rank, **launch_params(ws_raw, R))
            dp.set_stream(torch.cuda.current_stream().cuda_stream)
            dp.lenet_load(weights)
            dp.set_stage_timing(1)
            torch.cuda.synchronize()

            def plain_step_same_cloud():
                dp.set_cloud_device(cloud_dev.data_ptr(), cloud_dev.shape[0], 12)
                dp.compute_normals()
                return dp.detect(sample_idx=idx_fe, seed=args.seed, do_prune=True, want_all=False)[1]

            dt0, sc0 = timed(plain_step_same_cloud)
            dp.close()
            assert sc0 == sc, (sc0, sc)   # the same hypotheses scored with and without the front end
            out["with_front_end"] = {"value": sc / dt, "unit": "hypotheses/s", "ms_per_step": dt * 1e3,
                                     "raw_points": int(raw.shape[0]), "voxels": int(fe["n_vox"]),
                                     "headline_cloud_points_inside_workspace": int(inside.sum()),
                                     "scored_per_step": sc, "graph_replays": int(fi.graph_replays),
                                     "fallbacks": int(fi.fallbacks),
                                     "same_cloud_without_front_end_ms": dt0 * 1e3,
                                     "front_end_cost_ms": (dt - dt0) * 1e3,
                                     "minus_headline_ms_per_step": dt * 1e3 - elapsed / K * 1e3,
                                     "note": ("ag2_detect_frame_raw: filter + voxel grid + sub-sampling + detect in one "
                                              "captured sequence, on a raw cloud (2.55 points per voxel, shuffled) whose "
                                              "voxelisation is the headline cloud; front_end_cost_ms = against the plain "
                                              "step on the very cloud and samples the front end produced; "
                                              "minus_headline_ms_per_step = against the headline step (the samples differ: "
                                              "drawn on the device here)")}

            def front_step():
                fe["n_vox2"] = df.preprocess_cloud_device(raw_dev.data_ptr(), raw.shape[0], 12,
                                                          voxel_size=scene.VOXEL)
                ns = df.subsample_uniformly(S, seed=args.seed, want_indices=False)
                df.compute_normals()
                return df.detect(n_resident=ns, seed=args.seed, do_prune=True, want_all=False)[1]

            dt, sc = timed(front_step)
            out["with_front_end"]["separate_calls_ms_per_step"] = dt * 1e3
            df.close()
        # (3) two independent clouds in flight at once (a streaming deployment): two contexts, each on
        # its own HIP stream and driven by its own host thread (ctypes releases the GIL inside the
        # library).  Throughput of the pair; the latency of one cloud is the single-stream figure.
        import threading
        pair, streams = [], []
        for _ in range(2):
            st_ = torch.cuda.Stream()
            dk = capi.Detector(device=local_rank, **launch_params(ws, R))
            dk.set_stream(st_.cuda_stream)
            dk.set_stage_timing(1)
            dk.lenet_load(weights)
            pair.append(dk)
            streams.append(st_)
        torch.cuda.synchronize()
        reps2, got = 20, [0, 0]

        def run_stream(k):
            for _ in range(reps2):
                pair[k].set_cloud_device(xyz_dev.data_ptr(), xyz.shape[0], 12)
                pair[k].compute_normals()
                got[k] += pair[k].detect(sample_idx=idx, seed=args.seed, do_prune=True, want_all=False)[1]

        for k in range(2):  # warm both contexts (buffer growth, first-launch costs)
            reps2 = 2
            run_stream(k)
        torch.cuda.synchronize()
        reps2, got = 20, [0, 0]
        with no_gc():
            t1 = time.perf_counter()
            th = [threading.Thread(target=run_stream, args=(k,)) for k in range(2)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
        out["two_streams"] = {"value": sum(got) / dt2, "unit": "hypotheses/s",
                              "ms_per_cloud_throughput": dt2 / (2 * reps2) * 1e3,
                              "note": "two clouds in flight on two HIP streams; never `value`"}
        for dk in pair:
            dk.close()
        # (3b) the same, from ONE host thread: ag2_pipe (two contexts taken in turn, frames submitted
        # asynchronously), clouds handed over in HOST memory (page-locked staging + asynchronous DMA) -- the
        # headline cloud with its samples, then the raw cloud of (2) through the front end
        pipe = capi.Pipe(device=local_rank, depth=2, **launch_params(ws, R))
        pipe.lenet_load(weights)

        def run_pipe(submit, reps):
            submit()
            scored = 0
            for _ in range(reps - 1):
                submit()
                scored += pipe.wait()[1]
            scored += pipe.wait()[1]
            return scored

        legs_pipe = {}
        for name, sub in (("headline_cloud_resident", lambda: pipe.submit(sample_idx=idx, seed=args.seed, dptr=xyz_dev.data_ptr(),
                                                                          n=xyz.shape[0], stride=12)),
                          ("headline_cloud_from_host", lambda: pipe.submit(xyz, idx, seed=args.seed)),
                          ("raw_cloud_from_host", (lambda: pipe.submit_raw(raw, num_samples=S, sample_seed=args.seed,
                                                                           seed=args.seed, voxel_size=scene.VOXEL))
                           if voxelised else None)):
            if sub is None:
                continue
            run_pipe(sub, 8)   # both contexts: step by step, fixed shapes + capture, first replays
            torch.cuda.synchronize()
            with no_gc():
                t1 = time.perf_counter()
                reps_p = 40
                sc = run_pipe(sub, reps_p)
                dtp = (time.perf_counter() - t1) / reps_p
                t1 = time.perf_counter()   # latency of ONE cloud through the same pipe (nothing else in flight)
                for _ in range(10):
                    sub()
                    pipe.wait()
                lat1 = (time.perf_counter() - t1) / 10
            legs_pipe[name] = {"ms_per_cloud": dtp * 1e3, "value": sc / reps_p / dtp, "unit": "hypotheses/s",
                               "scored_per_cloud": sc / reps_p, "single_cloud_latency_ms": lat1 * 1e3}
        pipe.close()
        out["async_pipeline"] = dict(legs_pipe, depth=2,
                                     note=("ONE host thread, ag2_pipe: two frames in flight on two HIP streams, clouds "
                                           "copied from pageable host memory into page-locked staging and transferred by "
                                           "asynchronous DMA; never `value`"))
        # (4) the launch-file variant with grasp clustering (launch/file_detect_grasps.launch:45
        # min_inliers = 5: HandleSearch::findClusters between the threshold and the top-k)
        d.set_min_inliers(5)

        def cluster_step():
            d.set_cloud_device(xyz_dev.data_ptr(), xyz.shape[0], 12)
            d.compute_normals()
            return d.detect(sample_idx=idx, seed=args.seed, do_prune=True, want_all=False)[1]

        dt, sc = timed(cluster_step)
        d.set_min_inliers(0)
        out["with_min_inliers_5"] = {"value": sc / dt, "unit": "hypotheses/s", "ms_per_step": dt * 1e3,
                                     "note": "launch-file min_inliers=5: k_cluster between threshold and top-k; never `value`"}
        out["value_pcie_inclusive"] = out["pcie_inclusive"]["value"]
        if args.config == "cfg2" and not os.environ.get("AG2_BENCH_NO_SIDE_CONFIGS"):
            # (5) BASELINE.json configurations 3 and 5 beside the headline (never `value`): a few steps each
            out["cfg3"] = side_cfg3(args, weights, local_rank)
            lat5 = stream_legs(args.seed, 20, 3, host_leg=False)
            out["cfg5"] = {"p50_ms": lat5["graph"]["p50_ms"], "p99_ms": lat5["graph"]["p99_ms"],
                           "within_budget": lat5["within_budget"], "same_bytes_as_stepwise": lat5["same_bytes_as_stepwise"],
                           "budget_ms": FRAME_BUDGET_MS, "value": lat5["graph"]["scored_per_s"], "unit": "hypotheses/s",
                           "scored_per_frame": lat5["graph"]["scored_per_frame"],
                           "stepwise_p50_ms": lat5["stepwise"]["p50_ms"], "stepwise_p99_ms": lat5["stepwise"]["p99_ms"],
                           "plain_fixed_shape_p50_ms": lat5["plain_fixed_shape"]["p50_ms"],
                           "preprocessed_frames_p50_ms": lat5["graph_preprocessed_frames"]["p50_ms"],
                           "front_end_ms_p50": lat5["front_end_ms_p50"],
                           "frame_info": lat5["graph"]["frame_info"], "raw_points_per_frame": lat5["raw_points_per_frame"],
                           # every frame of the graph leg, and where a slow one spent its time (library clock
                           # around the submit and the wait half; the collector's share; polling fall-backs)
                           "max_ms": lat5["graph"]["max_ms"], "max_frame": lat5["graph"]["max_frame"],
                           "lat_ms": lat5["graph"]["lat_ms"], "host_split": lat5["graph"]["host_split"],
                           "gc_ms_in_frames": lat5["graph"]["gc_ms_in_frames"],
                           "gc_disabled_in_leg": lat5["graph"]["gc_disabled_in_leg"], "gc": lat5["gc"],
                           "stepwise_max_ms": lat5["stepwise"]["max_ms"],
                           "workload": ("cfg5: 20 RAW frames (~765 k points -> ~300 k voxels of 3 mm) of a drifting tabletop "
                                        "scene, 2 000 samples per frame, ag2_detect_frame_raw (filter + voxel grid + "
                                        "sub-sampling + detect in one hipGraph); latency of one call, raw cloud in HBM -> "
                                        "selected grasps on the host")}
        out["cpu_baseline"] = cpu_baseline(xyz, ws, idx, R, weights)
        # BASELINE.json configs[0] ("PR1 ref"): the launch file's own case -- num_samples = 500, ONE thread
        # (launch/file_detect_grasps.launch:18-19) -- on a 50 k-point scene (the reference bundles no .pcd)
        n1, s1, r1, _, k1 = CONFIGS["cfg1"]
        xyz1, ws1 = scene.make_scene(args.seed, n1, kind=k1)
        idx1 = scene.draw_samples(args.seed, xyz1.shape[0], s1)
        cb1 = cpu_baseline(xyz1, ws1, idx1, r1, weights, budget_s=6.0, threads=1)
        cb1["workload"] = f"cfg1: {xyz1.shape[0]}-pt voxelised tabletop cloud, num_samples={s1}, {r1} orientations, num_threads=1"
        d1 = capi.Detector(device=local_rank, **launch_params(ws1, r1))
        d1.set_stream(torch.cuda.current_stream().cuda_stream)
        d1.lenet_load(weights)
        d1.set_stage_timing(1)

        def cfg1_step():
            d1.set_cloud(xyz1)
            d1.compute_normals()
            return d1.detect(sample_idx=idx1, seed=1, do_prune=True, want_all=False)[1]

        dt, sc = timed(cfg1_step)
        d1.close()
        cb1["gpu_same_workload"] = {"value": sc / dt, "unit": "hypotheses/s", "ms_per_step": dt * 1e3,
                                    "note": "this library on the same cfg1 workload, cloud handed over in host memory"}
        out["cpu_baseline_1thread"] = cb1
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)
    d.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
