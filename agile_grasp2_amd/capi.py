"""ctypes binding of the product library libag2hip.so (C-ABI: include/ag2_c.h).

Harness side only (tests, bench.py, smoke()).  There is no fallback of any kind: if the HIP
library is missing, or no GPU is usable, loading / creating a context raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# AG2_LIB lets a tuning experiment point the harness at another build of the SAME library
LIB_PATH = os.environ.get("AG2_LIB") or os.path.join(_HERE, "csrc", "libag2hip.so")

SYMBOLS = [
    "ag2_abi_version", "ag2_default_params", "ag2_create", "ag2_destroy", "ag2_last_error",
    "ag2_set_stream", "ag2_set_cloud", "ag2_set_cloud_device", "ag2_compute_normals",
    "ag2_get_normals", "ag2_get_grid_perm", "ag2_local_frames", "ag2_generate_hypotheses",
    "ag2_hyp_points", "ag2_prune", "ag2_render_images", "ag2_render_images_from_points",
    "ag2_lenet_load", "ag2_lenet_forward", "ag2_detect", "ag2_export_candidates_device",
    "ag2_get_counters", "ag2_get_stage_times",
    "ag2_preprocess_cloud", "ag2_preprocess_cloud_device", "ag2_get_cloud", "ag2_subsample_uniformly",
    "ag2_find_clusters", "ag2_set_min_inliers", "ag2_set_grid_origin", "ag2_set_stage_timing",
    "ag2_export_candidates_compact_device", "ag2_hand_constants",
    "ag2_stream_configure", "ag2_detect_frame", "ag2_get_frame_info",
    "ag2_export_selected_compact_device", "ag2_merge_selected_device",
    "ag2_detect_frame_raw", "ag2_get_samples", "ag2_gather_begin", "ag2_gather_selected", "ag2_merge_gathered",
    "ag2_submit_frame", "ag2_submit_frame_raw", "ag2_wait_frame", "ag2_pipe_create", "ag2_pipe_destroy",
    "ag2_pipe_last_error", "ag2_pipe_context", "ag2_pipe_lenet_load", "ag2_pipe_submit", "ag2_pipe_submit_raw",
    "ag2_pipe_wait", "ag2_set_wait_mode", "ag2_get_wait_info",
]


class RetryStep(RuntimeError):
    """ag2_merge_* returned AG2_ERR_RETRY: nothing was merged; every rank repeats its step (detect, export, exchange,
    merge) -- each rank's next detect runs step by step and learns its shapes again."""


class Params(C.Structure):
    _fields_ = [
        ("finger_width", C.c_double), ("hand_outer_diameter", C.c_double),
        ("hand_depth", C.c_double), ("hand_height", C.c_double), ("init_bite", C.c_double),
        ("nn_radius_taubin", C.c_double), ("nn_radius_hands", C.c_double),
        ("normals_radius", C.c_double), ("grid_cell", C.c_double),
        ("num_orientations", C.c_int32), ("num_threads", C.c_int32), ("n_cams", C.c_int32),
        ("filter_half_grasps", C.c_int32),
        ("cam_origin", (C.c_double * 3) * 2),
        ("workspace", C.c_double * 6),
        ("min_aperture", C.c_double), ("max_aperture", C.c_double),
        ("min_score_diff", C.c_double),
        ("num_selected", C.c_int32), ("debug_flags", C.c_int32),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "n_points", "n_valid_points", "n_samples", "n_frames", "n_hypotheses", "n_pruned",
        "n_scored", "n_selected", "sum_k1", "sum_k2", "sum_kcrop", "sum_p", "n_overflow_samples",
        "list_points", "detect_one_trip", "detect_redone")]


class FrameInfo(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "frames", "graph_replays", "plain_runs", "stepwise_runs", "captures", "capture_failed",
        "capture_refused", "fallbacks", "max_points", "max_samples", "max_cells", "max_images", "graph_ready",
        "last_fallback")]


class WaitInfo(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("poll", "spin_us", "poll_fallbacks", "poll_yields", "last_submit_us",
                                          "last_wait_us")]


class Times(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "grid_ms", "normals_ms", "frames_ms", "sweep_ms", "compact_ms", "render_ms",
        "lenet_conv_ms", "lenet_fc_ms", "select_ms", "total_ms", "sweep_overflow_ms", "preprocess_ms")]


HYP_DTYPE = np.dtype([
    ("axis", "<f8", 3), ("approach", "<f8", 3), ("binormal", "<f8", 3),
    ("surface", "<f8", 3), ("bottom", "<f8", 3), ("top", "<f8", 3),
    ("width", "<f8"), ("score", "<f8"),
    ("sample_slot", "<i4"), ("orientation", "<i4"),
    ("half_antipodal", "u1"), ("full_antipodal", "u1"), ("reserved", "<u2"), ("n_points", "<i4"),
])
assert HYP_DTYPE.itemsize == 176

_lib = None


def load():
    """Load libag2hip.so.  Raises if the extension was not built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
            "(make -C agile_grasp2_amd/csrc). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.ag2_create.restype = C.c_void_p
    L.ag2_create.argtypes = [C.POINTER(Params), C.c_int]
    L.ag2_destroy.argtypes = [C.c_void_p]
    L.ag2_last_error.restype = C.c_char_p
    L.ag2_last_error.argtypes = [C.c_void_p]
    L.ag2_default_params.argtypes = [C.POINTER(Params)]
    L.ag2_pipe_create.restype = C.c_void_p
    L.ag2_pipe_create.argtypes = [C.POINTER(Params), C.c_int, C.c_int]
    L.ag2_pipe_destroy.argtypes = [C.c_void_p]
    L.ag2_pipe_last_error.restype = C.c_char_p
    L.ag2_pipe_last_error.argtypes = [C.c_void_p]
    L.ag2_pipe_context.restype = C.c_void_p
    L.ag2_pipe_context.argtypes = [C.c_void_p, C.c_int]
    _lib = L
    return L


def default_params(**kw) -> Params:
    p = Params()
    load().ag2_default_params(C.byref(p))
    return apply_params(p, **kw)


def hand_constants(params: Params | None = None, **kw):
    """finger_spacing[20], angles[R], depths[] the hand sweep derives from the parameters (host
    only: no device is touched)."""
    p = params if params is not None else default_params(**kw)
    fs, ang, dep = np.zeros(20), np.zeros(int(p.num_orientations)), np.zeros(32)
    nd = C.c_int32(0)
    rc = load().ag2_hand_constants(C.byref(p), _ptr(fs), _ptr(ang), _ptr(dep), C.byref(nd))
    if rc:
        raise RuntimeError(f"ag2_hand_constants rc={rc}")
    return fs, ang, dep[: nd.value].copy()


def apply_params(p, **kw):
    for k, v in kw.items():
        if k == "cam_origin":
            a = np.asarray(v, dtype=np.float64).reshape(-1, 3)
            for i in range(a.shape[0]):
                for j in range(3):
                    p.cam_origin[i][j] = float(a[i, j])
        elif k == "workspace":
            for i in range(6):
                p.workspace[i] = float(v[i])
        else:
            setattr(p, k, v)
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Detector:
    """One context on one GPU.  Method names follow the C-ABI."""

    def __init__(self, params: Params | None = None, device: int = 0, **kw):
        self.L = load()
        self.params = params if params is not None else default_params(**kw)
        h = self.L.ag2_create(C.byref(self.params), C.c_int(device))
        if not h:
            raise RuntimeError("ag2_create failed: no usable HIP device or bad parameters "
                               "(libag2hip.so has no CPU fallback)")
        self.h = C.c_void_p(h)
        self.n = 0
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.L.ag2_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc == -5:   # AG2_ERR_RETRY: every rank repeats the step (a rank's one-trip detect did not hold its shapes)
            raise RetryStep(f"libag2hip (rc={rc}): " + self.L.ag2_last_error(self.h).decode())
        if rc != 0:
            raise RuntimeError(f"libag2hip (rc={rc}): " + self.L.ag2_last_error(self.h).decode())

    def set_stream(self, stream_ptr: int):
        self._ck(self.L.ag2_set_stream(self.h, C.c_void_p(stream_ptr)))

    def set_cloud(self, xyz, cam_source=None, normals=None):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        assert xyz.ndim == 2 and xyz.shape[1] >= 3
        self.n = xyz.shape[0]
        ncam, cs = 1, None
        if cam_source is not None:
            cs = np.asfortranarray(np.asarray(cam_source, dtype=np.int32))
            ncam = cs.shape[0]
        nr = None
        if normals is not None:
            nr = np.asfortranarray(np.asarray(normals, dtype=np.float64))
            assert nr.shape == (3, self.n)
        self._ck(self.L.ag2_set_cloud(self.h, _ptr(xyz), C.c_size_t(self.n),
                                      C.c_size_t(xyz.strides[0] if self.n else 12), _ptr(cs),
                                      C.c_int(ncam), _ptr(nr)))

    def set_cloud_device(self, dptr: int, n: int, stride_bytes: int = 12):
        self.n = n
        self._ck(self.L.ag2_set_cloud_device(self.h, C.c_void_p(dptr), C.c_size_t(n),
                                             C.c_size_t(stride_bytes)))

    def preprocess_cloud(self, xyz, cam_source=None, normals=None, filter_workspace=True,
                         voxelize=True, voxel_size=0.003, flags=0):
        """GraspDetector::preprocessPointCloud steps 1-2 on the GPU; returns the processed size."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        n = xyz.shape[0]
        ncam, cs, nr = 1, None, None
        if cam_source is not None:
            cs = np.asfortranarray(np.asarray(cam_source, dtype=np.int32))
            ncam = cs.shape[0]
        if normals is not None:
            nr = np.asfortranarray(np.asarray(normals, dtype=np.float64))
        m = C.c_size_t(0)
        self._ck(self.L.ag2_preprocess_cloud(
            self.h, _ptr(xyz), C.c_size_t(n), C.c_size_t(xyz.strides[0] if n else 12), _ptr(cs),
            C.c_int(ncam), _ptr(nr), C.c_int(int(filter_workspace)), C.c_int(int(voxelize)),
            C.c_double(voxel_size), C.c_int(flags), C.byref(m)))
        self.n = m.value
        return self.n

    def preprocess_cloud_device(self, dptr: int, n: int, stride_bytes: int = 12,
                                filter_workspace=True, voxelize=True, voxel_size=0.003):
        m = C.c_size_t(0)
        self._ck(self.L.ag2_preprocess_cloud_device(
            self.h, C.c_void_p(dptr), C.c_size_t(n), C.c_size_t(stride_bytes),
            C.c_int(int(filter_workspace)), C.c_int(int(voxelize)), C.c_double(voxel_size), C.byref(m)))
        self.n = m.value
        return self.n

    def get_cloud(self):
        ncam = int(self.params.n_cams)
        xyz = np.zeros((self.n, 3), dtype=np.float32)
        cam = np.zeros((ncam, self.n), dtype=np.int32, order="F")
        m = C.c_size_t(0)
        self._ck(self.L.ag2_get_cloud(self.h, _ptr(xyz), _ptr(cam), C.c_size_t(self.n), C.byref(m)))
        assert m.value == self.n
        return xyz, cam

    def subsample_uniformly(self, num_samples, seed=0, want_indices=True):
        """CloudCamera::subsampleUniformly; the indices also stay resident for detect(n_resident=...)."""
        out = np.zeros(max(1, min(num_samples, self.n)), dtype=np.int32) if want_indices else None
        m = C.c_size_t(0)
        self._ck(self.L.ag2_subsample_uniformly(
            self.h, C.c_size_t(num_samples), C.c_uint64(seed), _ptr(out),
            C.c_size_t(out.shape[0] if want_indices else 0), C.byref(m)))
        return out[: m.value].copy() if want_indices else m.value

    def compute_normals(self):
        self._ck(self.L.ag2_compute_normals(self.h))

    def get_normals(self):
        out = np.zeros((3, self.n), dtype=np.float64, order="F")
        self._ck(self.L.ag2_get_normals(self.h, _ptr(out)))
        return out

    def get_grid_perm(self):
        out = np.zeros(max(self.n, 1), dtype=np.int32)
        nv = C.c_size_t(0)
        self._ck(self.L.ag2_get_grid_perm(self.h, _ptr(out), C.c_size_t(out.shape[0]), C.byref(nv)))
        return out[: nv.value].copy()

    @staticmethod
    def _samples(sample_idx, sample_xyz, n_resident=None):
        si = sx = None
        if n_resident is not None:  # indices left on the device by subsample_uniformly
            assert sample_idx is None and sample_xyz is None
            return None, None, int(n_resident)
        if sample_idx is not None:
            si = np.ascontiguousarray(sample_idx, dtype=np.int32)
            s = si.shape[0]
        else:
            sx = np.asfortranarray(np.asarray(sample_xyz, dtype=np.float64))
            assert sx.shape[0] == 3
            s = sx.shape[1]
        return si, sx, s

    def local_frames(self, sample_idx=None, sample_xyz=None, slot_base=0, seed=0):
        si, sx, s = self._samples(sample_idx, sample_xyz)
        fr = np.zeros((s, 12), dtype=np.float64)
        valid = np.zeros(s, dtype=np.int32)
        self._ck(self.L.ag2_local_frames(self.h, _ptr(si), _ptr(sx), C.c_size_t(s),
                                         C.c_uint64(slot_base), C.c_uint64(seed), _ptr(fr), _ptr(valid)))
        return fr, valid

    def generate_hypotheses(self, sample_idx=None, sample_xyz=None, slot_base=0, seed=0,
                            n_resident=None):
        si, sx, s = self._samples(sample_idx, sample_xyz, n_resident)
        cap = max(1, s * int(self.params.num_orientations))
        out = np.zeros(cap, dtype=HYP_DTYPE)
        n = C.c_size_t(0)
        self._ck(self.L.ag2_generate_hypotheses(self.h, _ptr(si), _ptr(sx), C.c_size_t(s),
                                                C.c_uint64(slot_base), C.c_uint64(seed), _ptr(out),
                                                C.c_size_t(cap), C.byref(n)))
        return out[: n.value].copy()

    def hyp_points(self, h, p):
        pts = np.zeros((3, p), dtype=np.float64, order="F")
        nrm = np.zeros((3, p), dtype=np.float64, order="F")
        self._ck(self.L.ag2_hyp_points(self.h, C.c_size_t(h), _ptr(pts), _ptr(nrm)))
        return pts, nrm

    def prune(self, n):
        keep = np.zeros(n, dtype=np.uint8)
        self._ck(self.L.ag2_prune(self.h, _ptr(keep), C.c_size_t(n)))
        return keep

    def render_images(self, first, count):
        out = np.zeros((count, 60, 60, 3), dtype=np.uint8)
        self._ck(self.L.ag2_render_images(self.h, C.c_size_t(first), C.c_size_t(count), _ptr(out)))
        return out

    def render_images_from_points(self, pts_list, nrm_list):
        n = len(pts_list)
        offs = np.zeros(n + 1, dtype=np.int64)
        for i, p in enumerate(pts_list):
            offs[i + 1] = offs[i] + np.asarray(p).shape[1]
        tot = int(offs[-1])
        pts = np.zeros((3, max(tot, 1)), dtype=np.float64, order="F")
        nrm = np.zeros((3, max(tot, 1)), dtype=np.float64, order="F")
        for i in range(n):
            pts[:, offs[i]:offs[i + 1]] = pts_list[i]
            nrm[:, offs[i]:offs[i + 1]] = nrm_list[i]
        out = np.zeros((n, 60, 60, 3), dtype=np.uint8)
        self._ck(self.L.ag2_render_images_from_points(self.h, C.c_size_t(n), _ptr(offs), _ptr(pts),
                                                      _ptr(nrm), _ptr(out)))
        return out

    def lenet_load(self, w):
        arrs = [np.ascontiguousarray(w[k], dtype=np.float32) for k in (
            "conv1_w", "conv1_b", "conv2_w", "conv2_b", "ip1_w", "ip1_b", "ip2_w", "ip2_b")]
        self._ck(self.L.ag2_lenet_load(self.h, *[_ptr(a) for a in arrs]))

    def lenet_forward(self, images):
        images = np.ascontiguousarray(images, dtype=np.uint8)
        n = images.shape[0]
        out = np.zeros((n, 2), dtype=np.float32)
        self._ck(self.L.ag2_lenet_forward(self.h, _ptr(images), C.c_size_t(n), _ptr(out)))
        return out

    def detect(self, sample_idx=None, sample_xyz=None, slot_base=0, seed=0, do_prune=True,
               want_all=True, n_resident=None, local_select=True):
        si, sx, s = self._samples(sample_idx, sample_xyz, n_resident)
        if not local_select:   # a multi-GPU rank: the merge (merge_selected_device) does the top-k
            ns, na = C.c_size_t(0), C.c_size_t(0)
            self._ck(self.L.ag2_detect(self.h, _ptr(si), _ptr(sx), C.c_size_t(s), C.c_uint64(slot_base),
                                       C.c_uint64(seed), C.c_int(1 if do_prune else 0), None, C.c_size_t(0),
                                       C.byref(ns), None, C.c_size_t(0), C.byref(na)))
            # (na is 0 when the call ran in one trip: the count is known after the merge -- counters().n_scored)
            return np.zeros(0, dtype=HYP_DTYPE), na.value
        cap = max(1, s * int(self.params.num_orientations))
        # the selection is at most num_selected records (when that is >= 0): a buffer of that size, kept
        # between calls -- a fresh zero-filled table of every slot per call is host time inside a step
        nsel = int(self.params.num_selected)
        cap_sel = cap if nsel < 0 else max(1, min(cap, nsel))
        if getattr(self, "_sel_buf", None) is None or len(self._sel_buf) < cap_sel:
            self._sel_buf = np.zeros(cap_sel, dtype=HYP_DTYPE)
        sel = self._sel_buf
        allh = np.zeros(cap, dtype=HYP_DTYPE) if want_all else None
        ns, na = C.c_size_t(0), C.c_size_t(0)
        self._ck(self.L.ag2_detect(self.h, _ptr(si), _ptr(sx), C.c_size_t(s), C.c_uint64(slot_base),
                                   C.c_uint64(seed), C.c_int(1 if do_prune else 0), _ptr(sel),
                                   C.c_size_t(cap_sel), C.byref(ns),
                                   _ptr(allh) if want_all else None,
                                   C.c_size_t(cap if want_all else 0), C.byref(na)))
        return sel[: ns.value].copy(), (allh[: na.value].copy() if want_all else na.value)

    def stream_configure(self, max_points=0, max_samples=0, use_graph=True):
        self._ck(self.L.ag2_stream_configure(self.h, C.c_size_t(max_points), C.c_size_t(max_samples),
                                             C.c_int(1 if use_graph else 0)))

    def detect_frame(self, xyz=None, sample_idx=None, seed=0, do_prune=True, dptr=None, n=None, stride=12):
        """One frame of a cloud stream (ag2_detect_frame).  xyz: (n, 3) float32 in host memory, or
        dptr / n / stride for a device-resident cloud.  Returns (selected records, n_scored)."""
        si = np.ascontiguousarray(sample_idx, dtype=np.int32)
        if dptr is None:
            xyz = np.ascontiguousarray(xyz, dtype=np.float32)
            n, stride, ptr, on_dev = xyz.shape[0], 12, _ptr(xyz), 0
        else:
            ptr, on_dev = C.c_void_p(dptr), 1
        self.n = int(n)
        cap = max(1, len(si) * int(self.params.num_orientations))
        nsel = int(self.params.num_selected)   # (a buffer of the selection's size, kept between calls: as in detect)
        cap = cap if nsel < 0 else max(1, min(cap, nsel))
        if getattr(self, "_sel_buf", None) is None or len(self._sel_buf) < cap:
            self._sel_buf = np.zeros(cap, dtype=HYP_DTYPE)
        sel = self._sel_buf
        ns, na = C.c_size_t(0), C.c_size_t(0)
        self._ck(self.L.ag2_detect_frame(self.h, ptr, C.c_int(on_dev), C.c_size_t(n), C.c_size_t(stride),
                                         _ptr(si), C.c_size_t(len(si)), C.c_uint64(seed),
                                         C.c_int(1 if do_prune else 0), _ptr(sel), C.c_size_t(cap),
                                         C.byref(ns), C.byref(na)))
        return sel[: ns.value].copy(), na.value

    def detect_frame_raw(self, xyz=None, num_samples=0, sample_seed=0, seed=0, do_prune=True, dptr=None, n=None,
                         stride=12, filter_workspace=True, voxel_size=0.003):
        """One frame of the RAW cloud (ag2_detect_frame_raw): workspace filter, voxel grid and uniform
        sub-sampling on the device in front of the frame.  Returns (selected, n_scored, n_voxels)."""
        if dptr is None:
            xyz = np.ascontiguousarray(xyz, dtype=np.float32)
            n, stride, ptr, on_dev = xyz.shape[0], 12, _ptr(xyz), 0
        else:
            ptr, on_dev = C.c_void_p(dptr), 1
        cap = max(1, int(num_samples) * int(self.params.num_orientations))
        nsel = int(self.params.num_selected)
        cap_sel = cap if nsel < 0 else max(1, min(cap, nsel))
        if getattr(self, "_sel_buf", None) is None or len(self._sel_buf) < cap_sel:
            self._sel_buf = np.zeros(cap_sel, dtype=HYP_DTYPE)
        sel = self._sel_buf
        ns, na, nv = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self._ck(self.L.ag2_detect_frame_raw(self.h, ptr, C.c_int(on_dev), C.c_size_t(n), C.c_size_t(stride),
                                             C.c_int(int(filter_workspace)), C.c_double(voxel_size),
                                             C.c_size_t(num_samples), C.c_uint64(sample_seed), C.c_uint64(seed),
                                             C.c_int(1 if do_prune else 0), _ptr(sel), C.c_size_t(cap_sel),
                                             C.byref(ns), C.byref(na), C.byref(nv)))
        self.n = int(nv.value)
        return sel[: ns.value].copy(), na.value, nv.value

    def get_samples(self):
        """The sample indices the device drew last (subsample_uniformly / detect_frame_raw)."""
        m = C.c_size_t(0)
        cap = max(1, self.n)
        out = np.zeros(cap, dtype=np.int32)
        self._ck(self.L.ag2_get_samples(self.h, _ptr(out), C.c_size_t(cap), C.byref(m)))
        return out[: m.value].copy()

    def frame_info(self) -> FrameInfo:
        fi = FrameInfo()
        self._ck(self.L.ag2_get_frame_info(self.h, C.byref(fi)))
        return fi

    def set_min_inliers(self, k: int):
        """HandleSearch::setMinInliers: > 0 makes detect() cluster before the top-k."""
        self._ck(self.L.ag2_set_min_inliers(self.h, C.c_int(k)))

    def set_grid_origin(self, origin):
        """Spatial tiles: bin (and prune) against the whole cloud's minimum; None = automatic."""
        o = None if origin is None else np.ascontiguousarray(origin, dtype=np.float32)
        assert o is None or o.shape == (3,)
        self._ck(self.L.ag2_set_grid_origin(self.h, _ptr(o)))

    def set_stage_timing(self, level: int):
        """2: events around every stage (default), 1: around the sweep only, 0: none."""
        self._ck(self.L.ag2_set_stage_timing(self.h, C.c_int(level)))

    def find_clusters(self, hands, min_inliers: int):
        """HandleSearch::findClusters(hand_list) on the GPU; hands: HYP_DTYPE array."""
        hands = np.ascontiguousarray(hands, dtype=HYP_DTYPE)
        out = np.zeros(max(1, len(hands)), dtype=HYP_DTYPE)
        n = C.c_size_t(0)
        self._ck(self.L.ag2_find_clusters(self.h, _ptr(hands), C.c_size_t(len(hands)),
                                          C.c_int(min_inliers), _ptr(out), C.c_size_t(len(out)),
                                          C.byref(n)))
        return out[: n.value].copy()

    def export_candidates_device(self, dptr: int, nbytes: int):
        self._ck(self.L.ag2_export_candidates_device(self.h, C.c_void_p(dptr), C.c_size_t(nbytes)))

    def export_candidates_compact_device(self, dptr: int, nbytes: int, cap_records: int):
        self._ck(self.L.ag2_export_candidates_compact_device(self.h, C.c_void_p(dptr), C.c_size_t(nbytes),
                                                             C.c_size_t(cap_records)))

    def export_selected_compact_device(self, dptr: int, nbytes: int, cap_records: int):
        self._ck(self.L.ag2_export_selected_compact_device(self.h, C.c_void_p(dptr), C.c_size_t(nbytes),
                                                           C.c_size_t(cap_records)))

    def merge_selected_device(self, dptr: int, world: int, cap_records: int):
        """(top num_selected of the gathered selected lists, number of records that took part)"""
        k = int(self.params.num_selected)
        cap = world * cap_records if k < 0 else min(k, world * cap_records)
        sel = np.zeros(max(1, cap), dtype=HYP_DTYPE)
        ns, nt = C.c_size_t(0), C.c_size_t(0)
        self._ck(self.L.ag2_merge_selected_device(self.h, C.c_void_p(dptr), C.c_size_t(world),
                                                  C.c_size_t(cap_records), _ptr(sel), C.c_size_t(len(sel)),
                                                  C.byref(ns), C.byref(nt)))
        return sel[: ns.value].copy(), nt.value

    def set_wait_mode(self, poll: bool = True, spin_us: int = 50):
        """How the host waits for results: poll the flag behind them (spin spin_us, then yield), or the stream."""
        self._ck(self.L.ag2_set_wait_mode(self.h, C.c_int(1 if poll else 0), C.c_int(int(spin_us))))

    def wait_info(self) -> WaitInfo:
        w = WaitInfo()
        self._ck(self.L.ag2_get_wait_info(self.h, C.byref(w)))
        return w

    def counters(self) -> Counters:
        c = Counters()
        self._ck(self.L.ag2_get_counters(self.h, C.byref(c)))
        return c

    def times(self) -> Times:
        t = Times()
        self._ck(self.L.ag2_get_stage_times(self.h, C.byref(t)))
        return t


class Pipe:
    """ag2_pipe: `depth` contexts on one GPU taken in turn by one caller thread (frames in flight overlap)."""

    def __init__(self, params: Params | None = None, device: int = 0, depth: int = 2, **kw):
        self.L = load()
        self.params = params if params is not None else default_params(**kw)
        h = self.L.ag2_pipe_create(C.byref(self.params), C.c_int(device), C.c_int(depth))
        if not h:
            raise RuntimeError("ag2_pipe_create failed: no usable HIP device or bad parameters")
        self.h = C.c_void_p(h)
        self.depth = depth
        self._caps = []   # one result capacity per frame in flight, oldest first (frames may differ in num_samples)

    def close(self):
        if getattr(self, "h", None):
            self.L.ag2_pipe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(f"libag2hip pipe (rc={rc}): " + self.L.ag2_pipe_last_error(self.h).decode())

    def lenet_load(self, w):
        arrs = [np.ascontiguousarray(w[k], dtype=np.float32) for k in (
            "conv1_w", "conv1_b", "conv2_w", "conv2_b", "ip1_w", "ip1_b", "ip2_w", "ip2_b")]
        self._ck(self.L.ag2_pipe_lenet_load(self.h, *[_ptr(a) for a in arrs]))

    def stream_configure(self, use_graph=True):
        for k in range(self.depth):
            c = C.c_void_p(self.L.ag2_pipe_context(self.h, C.c_int(k)))
            rc = self.L.ag2_stream_configure(c, C.c_size_t(0), C.c_size_t(0), C.c_int(1 if use_graph else 0))
            if rc:
                raise RuntimeError(f"ag2_stream_configure rc={rc}")

    def submit_raw(self, xyz=None, num_samples=0, sample_seed=0, seed=0, do_prune=True, dptr=None, n=None, stride=12,
                   filter_workspace=True, voxel_size=0.003):
        if dptr is None:
            xyz = np.ascontiguousarray(xyz, dtype=np.float32)
            n, stride, ptr, on_dev = xyz.shape[0], 12, _ptr(xyz), 0
        else:
            ptr, on_dev = C.c_void_p(dptr), 1
        cap = max(1, int(num_samples) * int(self.params.num_orientations))
        self._ck(self.L.ag2_pipe_submit_raw(self.h, ptr, C.c_int(on_dev), C.c_size_t(n), C.c_size_t(stride),
                                            C.c_int(int(filter_workspace)), C.c_double(voxel_size),
                                            C.c_size_t(num_samples), C.c_uint64(sample_seed), C.c_uint64(seed),
                                            C.c_int(1 if do_prune else 0)))
        self._caps.append(cap)

    def submit(self, xyz=None, sample_idx=None, seed=0, do_prune=True, dptr=None, n=None, stride=12):
        si = np.ascontiguousarray(sample_idx, dtype=np.int32)
        if dptr is None:
            xyz = np.ascontiguousarray(xyz, dtype=np.float32)
            n, stride, ptr, on_dev = xyz.shape[0], 12, _ptr(xyz), 0
        else:
            ptr, on_dev = C.c_void_p(dptr), 1
        cap = max(1, len(si) * int(self.params.num_orientations))
        self._ck(self.L.ag2_pipe_submit(self.h, ptr, C.c_int(on_dev), C.c_size_t(n), C.c_size_t(stride), _ptr(si),
                                        C.c_size_t(len(si)), C.c_uint64(seed), C.c_int(1 if do_prune else 0)))
        self._caps.append(cap)

    def wait(self):
        """(selected, n_scored, n_voxels) of the oldest frame in flight"""
        nsel = int(self.params.num_selected)
        cap0 = self._caps[0] if self._caps else 1   # the capacity of the OLDEST frame, not of the last submit
        cap = cap0 if nsel < 0 else max(1, min(cap0, nsel))
        if getattr(self, "_sel_buf", None) is None or len(self._sel_buf) < cap:
            self._sel_buf = np.zeros(cap, dtype=HYP_DTYPE)
        ns, na, nv = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        rc = self.L.ag2_pipe_wait(self.h, _ptr(self._sel_buf), C.c_size_t(len(self._sel_buf)), C.byref(ns), C.byref(na),
                                  C.byref(nv))
        if rc == -3 and ns.value > len(self._sel_buf):   # AG2_ERR_CAPACITY: the frame is kept, wait again with room
            self._sel_buf = np.zeros(ns.value, dtype=HYP_DTYPE)
            rc = self.L.ag2_pipe_wait(self.h, _ptr(self._sel_buf), C.c_size_t(len(self._sel_buf)), C.byref(ns),
                                      C.byref(na), C.byref(nv))
        self._ck(rc)
        if self._caps:
            self._caps.pop(0)
        return self._sel_buf[: ns.value].copy(), na.value, nv.value
