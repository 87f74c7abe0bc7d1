"""ctypes binding of the CPU oracle (oracle/libag2_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package agile_grasp2_amd.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libag2_oracle.so")


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
        ("num_selected", C.c_int32), ("reserved", C.c_int32),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "n_points", "n_valid_points", "n_samples", "n_frames", "n_hypotheses", "n_pruned",
        "n_scored", "n_selected", "sum_k1", "sum_k2", "sum_kcrop", "sum_p")] + [
        (n, C.c_double) for n in ("t_normals", "t_frames", "t_hands", "t_images", "t_lenet", "t_total")]


HYP_DTYPE = np.dtype([
    ("axis", "<f8", 3), ("approach", "<f8", 3), ("binormal", "<f8", 3),
    ("surface", "<f8", 3), ("bottom", "<f8", 3), ("top", "<f8", 3),
    ("width", "<f8"), ("score", "<f8"),
    ("sample_slot", "<i4"), ("orientation", "<i4"),
    ("half_antipodal", "u1"), ("full_antipodal", "u1"), ("reserved", "<u2"), ("n_points", "<i4"),
])
assert HYP_DTYPE.itemsize == 176


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "ag2_oracle.cpp"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.ag2o_create.restype = C.c_void_p
        L.ag2o_create.argtypes = [C.POINTER(Params)]
        L.ag2o_destroy.argtypes = [C.c_void_p]
        L.ag2o_last_error.restype = C.c_char_p
        L.ag2o_last_error.argtypes = [C.c_void_p]
        L.ag2o_default_params.argtypes = [C.POINTER(Params)]
        _lib = L
    return _lib


def default_params(**kw) -> Params:
    p = Params()
    lib().ag2o_default_params(C.byref(p))
    apply_params(p, **kw)
    return p


def hand_constants(params: Params | None = None, **kw):
    p = params if params is not None else default_params(**kw)
    fs, ang, dep = np.zeros(20), np.zeros(int(p.num_orientations)), np.zeros(32)
    nd = C.c_int32(0)
    rc = lib().ag2o_hand_constants(C.byref(p), _ptr(fs), _ptr(ang), _ptr(dep), C.byref(nd))
    assert rc == 0
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


def _ptr(a, t=C.c_void_p):
    return None if a is None else a.ctypes.data_as(t)


class Oracle:
    """Thin object wrapper; method names mirror the product's C-ABI (include/ag2_c.h)."""

    def __init__(self, params: Params | None = None, **kw):
        self.L = lib()
        self.params = params if params is not None else default_params(**kw)
        self.h = C.c_void_p(self.L.ag2o_create(C.byref(self.params)))
        self.n = 0
        self._keep = []

    def close(self):
        if self.h:
            self.L.ag2o_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError("oracle: " + self.L.ag2o_last_error(self.h).decode())

    def set_cloud(self, xyz, cam_source=None, normals=None):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        assert xyz.ndim == 2 and xyz.shape[1] >= 3
        self.n = xyz.shape[0]
        ncam = 1
        self.n_cams = 1 if cam_source is None else np.asarray(cam_source).shape[0]
        cs = None
        if cam_source is not None:
            cs = np.asarray(cam_source, dtype=np.int32)
            ncam = cs.shape[0]
            cs = np.asfortranarray(cs)
        nr = None
        if normals is not None:
            nr = np.asfortranarray(np.asarray(normals, dtype=np.float64))
            assert nr.shape == (3, self.n)
        self._ck(self.L.ag2o_set_cloud(self.h, _ptr(xyz), C.c_size_t(self.n),
                                       C.c_size_t(xyz.strides[0]), _ptr(cs), C.c_int(ncam), _ptr(nr)))

    def preprocess_cloud(self, xyz, cam_source=None, normals=None, filter_workspace=True,
                         voxelize=True, voxel_size=0.003, flags=0):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        n = xyz.shape[0]
        ncam, cs, nr = 1, None, None
        if cam_source is not None:
            cs = np.asfortranarray(np.asarray(cam_source, dtype=np.int32))
            ncam = cs.shape[0]
        if normals is not None:
            nr = np.asfortranarray(np.asarray(normals, dtype=np.float64))
        m = C.c_size_t(0)
        stride = xyz.strides[0] if n else 12
        self._ck(self.L.ag2o_preprocess_cloud(
            self.h, _ptr(xyz), C.c_size_t(n), C.c_size_t(stride), _ptr(cs), C.c_int(ncam), _ptr(nr),
            C.c_int(int(filter_workspace)), C.c_int(int(voxelize)), C.c_double(voxel_size),
            C.c_int(flags), C.byref(m)))
        self.n = m.value
        self.n_cams = ncam
        return self.n

    def get_cloud(self):
        ncam = getattr(self, "n_cams", 1)
        xyz = np.zeros((self.n, 3), dtype=np.float32)
        cam = np.zeros((ncam, self.n), dtype=np.int32, order="F")
        m = C.c_size_t(0)
        self._ck(self.L.ag2o_get_cloud(self.h, _ptr(xyz), _ptr(cam), C.c_size_t(self.n), C.byref(m)))
        return xyz, cam

    def subsample_uniformly(self, num_samples, seed=0):
        out = np.zeros(max(1, min(num_samples, self.n)), dtype=np.int32)
        m = C.c_size_t(0)
        self._ck(self.L.ag2o_subsample_uniformly(self.h, C.c_size_t(num_samples), C.c_uint64(seed),
                                                 _ptr(out), C.c_size_t(out.shape[0]), C.byref(m)))
        return out[: m.value].copy()

    def compute_normals(self):
        self._ck(self.L.ag2o_compute_normals(self.h))

    def get_normals(self):
        out = np.zeros((3, self.n), dtype=np.float64, order="F")
        self._ck(self.L.ag2o_get_normals(self.h, _ptr(out)))
        return out

    def get_grid_perm(self):
        out = np.zeros(self.n, dtype=np.int32)
        self._ck(self.L.ag2o_get_grid_perm(self.h, _ptr(out)))
        return out

    def radius_search(self, q, r, cap=1 << 20):
        q = np.ascontiguousarray(q, dtype=np.float32)
        out = np.zeros(cap, dtype=np.int32)
        n = C.c_size_t(0)
        self._ck(self.L.ag2o_radius_search(self.h, _ptr(q), C.c_double(r), _ptr(out),
                                           C.c_size_t(cap), C.byref(n)))
        return out[: n.value].copy()

    @staticmethod
    def _samples(sample_idx, sample_xyz):
        si = sx = None
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
        self._ck(self.L.ag2o_local_frames(self.h, _ptr(si), _ptr(sx), C.c_size_t(s),
                                          C.c_uint64(slot_base), C.c_uint64(seed), _ptr(fr), _ptr(valid)))
        return fr, valid

    def generate_hypotheses(self, sample_idx=None, sample_xyz=None, slot_base=0, seed=0):
        si, sx, s = self._samples(sample_idx, sample_xyz)
        cap = max(1, s * int(self.params.num_orientations))
        out = np.zeros(cap, dtype=HYP_DTYPE)
        n = C.c_size_t(0)
        self._ck(self.L.ag2o_generate_hypotheses(self.h, _ptr(si), _ptr(sx), C.c_size_t(s),
                                                 C.c_uint64(slot_base), C.c_uint64(seed), _ptr(out),
                                                 C.c_size_t(cap), C.byref(n)))
        return out[: n.value].copy()

    def hyp_points(self, h, p):
        pts = np.zeros((3, p), dtype=np.float64, order="F")
        nrm = np.zeros((3, p), dtype=np.float64, order="F")
        self._ck(self.L.ag2o_hyp_points(self.h, C.c_size_t(h), _ptr(pts), _ptr(nrm)))
        return pts, nrm

    def prune(self, n):
        keep = np.zeros(n, dtype=np.uint8)
        self._ck(self.L.ag2o_prune(self.h, _ptr(keep), C.c_size_t(n)))
        return keep

    def render_images(self, first, count):
        out = np.zeros((count, 60, 60, 3), dtype=np.uint8)
        self._ck(self.L.ag2o_render_images(self.h, C.c_size_t(first), C.c_size_t(count), _ptr(out)))
        return out

    def render_image_from_points(self, pts, nrm):
        pts = np.asfortranarray(np.asarray(pts, dtype=np.float64))
        nrm = np.asfortranarray(np.asarray(nrm, dtype=np.float64))
        out = np.zeros((60, 60, 3), dtype=np.uint8)
        self.L.ag2o_render_image_from_points(_ptr(pts), _ptr(nrm), C.c_size_t(pts.shape[1]), _ptr(out))
        return out

    def lenet_load(self, w):
        arrs = [np.ascontiguousarray(w[k], dtype=np.float32) for k in (
            "conv1_w", "conv1_b", "conv2_w", "conv2_b", "ip1_w", "ip1_b", "ip2_w", "ip2_b")]
        self._keep = arrs
        self._ck(self.L.ag2o_lenet_load(self.h, *[_ptr(a) for a in arrs]))

    def lenet_forward(self, images):
        images = np.ascontiguousarray(images, dtype=np.uint8)
        n = images.shape[0]
        out = np.zeros((n, 2), dtype=np.float32)
        self._ck(self.L.ag2o_lenet_forward(self.h, _ptr(images), C.c_size_t(n), _ptr(out)))
        return out

    def detect(self, sample_idx=None, sample_xyz=None, slot_base=0, seed=0, do_prune=True):
        si, sx, s = self._samples(sample_idx, sample_xyz)
        cap = max(1, s * int(self.params.num_orientations))
        sel = np.zeros(cap, dtype=HYP_DTYPE)
        allh = np.zeros(cap, dtype=HYP_DTYPE)
        ns, na = C.c_size_t(0), C.c_size_t(0)
        self._ck(self.L.ag2o_detect(self.h, _ptr(si), _ptr(sx), C.c_size_t(s), C.c_uint64(slot_base),
                                    C.c_uint64(seed), C.c_int(1 if do_prune else 0), _ptr(sel),
                                    C.c_size_t(cap), C.byref(ns), _ptr(allh), C.c_size_t(cap),
                                    C.byref(na)))
        return sel[: ns.value].copy(), allh[: na.value].copy()

    def set_grid_origin(self, origin):
        o = None if origin is None else np.ascontiguousarray(origin, dtype=np.float32)
        self._ck(self.L.ag2o_set_grid_origin(self.h, _ptr(o)))

    def set_min_inliers(self, k):
        self._ck(self.L.ag2o_set_min_inliers(self.h, C.c_int(k)))

    def find_clusters(self, hands, min_inliers, remove_inliers=False):
        hands = np.ascontiguousarray(hands, dtype=HYP_DTYPE)
        out = np.zeros(max(1, len(hands)), dtype=HYP_DTYPE)
        n = C.c_size_t(0)
        rc = self.L.ag2o_find_clusters(_ptr(hands), C.c_size_t(len(hands)), C.c_int(min_inliers),
                                       C.c_int(int(remove_inliers)), _ptr(out), C.c_size_t(len(out)),
                                       C.byref(n))
        if rc != 0:
            raise RuntimeError(f"oracle: find_clusters rc={rc}")
        return out[: n.value].copy()

    def counters(self) -> Counters:
        c = Counters()
        self._ck(self.L.ag2o_get_counters(self.h, C.byref(c)))
        return c
