"""Independent numpy/scipy re-derivations used to pin the C++ oracle (tests only).

Each function recomputes one stage of the reference algorithm a second way (brute force, LAPACK
eigh, scipy.ndimage, torch-CPU conv) so that a mistake in oracle/ag2_oracle.cpp does not silently
become the definition of "correct".  Citations are to /root/reference files.
"""
from __future__ import annotations

import numpy as np

MASK64 = (1 << 64) - 1


def draw_u64(seed: int, slot: int, j: int) -> int:
    """The counter RNG that replaces rand() (hand_search.cpp:130), in Python integers."""
    x = (seed ^ ((0x9E3779B97F4A7C15 * (slot + 1)) & MASK64)) & MASK64
    x = (x + ((0xD1B54A32D192ED03 * (j + 1)) & MASK64)) & MASK64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & MASK64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & MASK64
    x ^= x >> 31
    return x


def grid_keys(xyz: np.ndarray, cell: float = 0.01):
    """Cell key of every point: float32 arithmetic, x fastest."""
    xyz = xyz.astype(np.float32)
    o = xyz.min(axis=0)
    inv = np.float32(1.0) / np.float32(cell)
    c = np.floor((xyz - o) * inv).astype(np.int64)
    dims = c.max(axis=0) + 1
    return (c[:, 2] * dims[1] + c[:, 1]) * dims[0] + c[:, 0]


def canonical_order(xyz: np.ndarray, cell: float = 0.01) -> np.ndarray:
    """sorted position -> original index, ascending (cell key, index)."""
    k = grid_keys(xyz, cell)
    return np.lexsort((np.arange(len(k)), k)).astype(np.int32)


def brute_radius(xyz: np.ndarray, q: np.ndarray, r: float, rank: np.ndarray) -> np.ndarray:
    """Exact radius search semantics of pcl::KdTreeFLANN::radiusSearch (hand_search.cpp:122,:201):
    float squared distance strictly below (float)(r*r); returned in canonical order."""
    xyz = xyz.astype(np.float32)
    d = xyz - q.astype(np.float32)[None, :]
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    idx = np.nonzero(d2 < np.float32(r * r))[0]
    return idx[np.argsort(rank[idx], kind="stable")].astype(np.int32)


def pca_normal(nb_xyz: np.ndarray, p: np.ndarray) -> np.ndarray:
    """Plane-fit normal (pcl::NormalEstimationOMP, hand_search.cpp:85-92) in float64 with LAPACK,
    flipped towards the viewpoint (0,0,0)."""
    a = nb_xyz.astype(np.float64)
    c = np.cov(a.T, bias=True)
    w, v = np.linalg.eigh(c)
    n = v[:, 0]
    if np.dot(-p.astype(np.float64), n) < 0:
        n = -n
    return n, w


def local_frame(normals_nb: np.ndarray, sample: np.ndarray, cam_origin: np.ndarray,
                seed: int, slot: int):
    """LocalFrame::findAverageNormalAxis (local_frame.cpp:26-59) on the drawn normals.
    normals_nb: K x 3 finite neighbour normals in canonical order."""
    k = normals_nb.shape[0]
    m = min(50, k)
    picks = [draw_u64(seed, slot, j) % k for j in range(m)]
    N = normals_nb[picks].astype(np.float64)
    N = N / np.linalg.norm(N, axis=1, keepdims=True)
    M = N.T @ N
    w, v = np.linalg.eigh(M)
    c = v[:, 0]
    G = (N @ N.T) ** 6
    jmax = int(np.argmax(G.sum(axis=0)))
    npart = (np.eye(3) - np.outer(c, c)) @ N[jmax]
    normal = npart / np.linalg.norm(npart)
    binormal = np.cross(c, normal)
    v2 = sample.astype(np.float64) - cam_origin
    if normal @ v2 > 0:
        normal = -normal
    if binormal @ v2 > 0:
        binormal = -binormal
    curv = np.cross(normal, binormal)
    return normal, binormal, curv, w


def finger_tables(od=0.09, fw=0.01):
    """FingerHand ctor, finger_hand.cpp:7-12."""
    fs_half = np.array([0.0 + i * ((od - fw) / 9.0) for i in range(10)])
    fs = np.concatenate([(fs_half - od) + fw, fs_half])
    return fs, fs + fw


def sweep_sample(P, Q, frame, sample, prm):
    """HandSearch::calculateHand (hand_search.cpp:319-426) for one sample, vectorised numpy.
    P, Q: K2 x 3 float64 centred points / normals (canonical order); frame: 3x3 [n b c] columns.
    Returns a list of dicts."""
    fw, od, depth, hh, bite, R = (prm["finger_width"], prm["hand_outer_diameter"], prm["hand_depth"],
                                  prm["hand_height"], prm["init_bite"], prm["num_orientations"])
    fs, fsr = finger_tables(od, fw)
    z = P @ frame[:, 2]
    keep = (z > -hh) & (z < hh)
    Pc, Qc = P[keep], Q[keep]
    out = []
    if Pc.shape[0] == 0:
        return out
    depths = []
    d = bite + 0.005
    while d <= depth:
        depths.append(d)
        d += 0.005
    for oi in range(R):
        a = -np.pi / 2 + oi * (np.pi / R)
        rot = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0, 0, 1.0]])
        Fr = frame @ rot
        X = Pc @ Fr
        Y = Qc @ Fr
        top, bottom = bite, bite - depth
        cset = X[:, 1] < top
        if not cset.any() or (X[cset, 1] < bottom).any():
            continue
        xc = X[cset, 0]
        free = np.array([not ((xc > fs[k]) & (xc < fsr[k])).any() for k in range(20)])
        if free.sum() <= 2:
            continue
        hand = free[:10] & free[10:]
        if hand.sum() == 0:
            continue
        valid = np.nonzero(hand)[0]
        idx = int(valid[int(np.ceil(len(valid) / 2.0)) - 1])
        for dd in depths:
            cs = X[:, 1] < dd
            if (X[cs, 1] < dd - depth).any():
                break
            xs = X[cs, 0]
            if ((xs > fs[idx]) & (xs < fsr[idx])).any() or ((xs > fs[10 + idx]) & (xs < fsr[10 + idx])).any():
                break
            top, bottom = dd, dd - depth
        left, right = fs[idx] + fw, fs[10 + idx]
        center = 0.5 * (left + right)
        surface = X[:, 1].min()
        box = (X[:, 1] < top) & (X[:, 0] > left) & (X[:, 0] < right)
        if not box.any():
            continue
        XB, YB = X[box], Y[box]
        width = XB[:, 0].max() - XB[:, 0].min()
        lc = left - 0.5 * (0.1 - (right - left))
        U = np.stack([(1.0 / 0.1) * (XB[:, 0] - lc), (1.0 / (top - bottom)) * (XB[:, 1] - bottom),
                      (1.0 / (2.0 * hh)) * (XB[:, 2] + hh)], axis=1)
        cosf = np.cos(30.0 * np.pi / 180.0)
        le = U[:, 0] < U[:, 0].min() + 0.003
        re = U[:, 0] > U[:, 0].max() - 0.003
        lv = le & (-YB[:, 0] > cosf)
        rv = re & (YB[:, 0] > cosf)
        label = 0
        if lv.any() or rv.any():
            label = 1
        if lv.any() and rv.any():
            ty = min(U[lv, 1].max(), U[rv, 1].max())
            by = max(U[lv, 1].min(), U[rv, 1].min())
            tz = min(U[lv, 2].max(), U[rv, 2].max())
            bz = max(U[lv, 2].min(), U[rv, 2].min())
            if ty > by and tz > bz:
                label = 2
        out.append(dict(
            orientation=oi, binormal=Fr[:, 0], approach=Fr[:, 1], axis=Fr[:, 2],
            surface=Fr @ np.array([center, surface, 0.0]) + sample,
            bottom=Fr @ np.array([center, bottom, 0.0]) + sample,
            top=Fr @ np.array([center, top, 0.0]) + sample,
            width=width, label=label, U=U, Y=YB))
    return out


def render_image(U: np.ndarray, Y: np.ndarray) -> np.ndarray:
    """Learning::convertToImageRGB + convertTo(CV_8UC3,255) (learning.cpp:143-209, :16) with
    scipy's maximum_filter as the 3x3 dilate.  U, Y: P x 3."""
    from scipy.ndimage import maximum_filter
    S = 60
    y = U[:, 1] - U[:, 1].min()
    cs = 1.0 / S
    cell = np.floor(U[:, 0] / cs).astype(np.int64) + np.floor(y / cs).astype(np.int64) * S
    ok = (cell >= 0) & (cell < S * S)
    acc = np.zeros((S * S, 3))
    np.add.at(acc, cell[ok], Y[ok])  # sequential, in order
    cnt = np.bincount(cell[ok], minlength=S * S)
    img = np.zeros((S, S, 3), dtype=np.float32)
    for c in np.nonzero(cnt)[0]:
        a = acc[c]
        with np.errstate(all="ignore"):
            v = np.abs((1.0 / np.sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2])) * a)
        v = np.where(np.isnan(v), 0.0, v)
        img[S - 1 - c // S, c % S] = v.astype(np.float32)
    dil = maximum_filter(img, size=(3, 3, 1), mode="constant", cval=0.0)
    rgb = dil[:, :, ::-1]
    t = rgb.astype(np.float32) * np.float32(255.0)
    return np.clip(np.rint(t), 0, 255).astype(np.uint8)


def lenet_torch(w: dict, images_hwc: np.ndarray) -> np.ndarray:
    """Caffe LeNet of caffe/test_1batch2.prototxt via torch CPU ops (fp32)."""
    import torch
    import torch.nn.functional as F
    x = torch.from_numpy(images_hwc.astype(np.float32)).permute(0, 3, 1, 2).contiguous()
    t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}
    with torch.no_grad():
        x = F.max_pool2d(F.conv2d(x, t["conv1_w"], t["conv1_b"]), 2)
        x = F.max_pool2d(F.conv2d(x, t["conv2_w"], t["conv2_b"]), 2)
        x = F.relu(F.linear(x.flatten(1), t["ip1_w"], t["ip1_b"]))
        x = F.linear(x, t["ip2_w"], t["ip2_b"])
    return x.numpy()
