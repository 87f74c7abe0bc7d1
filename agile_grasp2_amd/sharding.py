"""Multi-GPU sharding of the hot path: one process per GPU, samples sharded by range, one
all-gather of the fixed-slot candidate table (RCCL over xGMI when the backend is "nccl").

After the normals every sample is independent (src/agile_grasp2/hand_search.cpp:194-219 keeps no
cross-iteration state; results are concatenated in sample order, :223-228), so the only exchange
step is the gather of the per-sample results.  Each rank fills the slots
(global_sample * R + orientation) of its own sample range in a table of 176-byte records
(ag2_hypothesis, n_points == 0 marks an empty slot); concatenating the rank tables in rank order is
the reference's output order, whatever the number of ranks.  The neighbour-draw RNG is keyed by the
GLOBAL sample slot (slot_base + i), so results do not depend on the sharding.

Harness-side plumbing (torch.distributed), shared by bench.py and the gloo tests.
"""
from __future__ import annotations

import numpy as np

SLOT_BYTES = 176


def shard_range(n_samples: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced [begin, end) of the sample list owned by `rank`."""
    base, rem = divmod(n_samples, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def max_shard(n_samples: int, world: int) -> int:
    return (n_samples + world - 1) // world


def table_from_records(records: np.ndarray, slot_base: int, n_local_samples: int, R: int,
                       pad_samples: int) -> np.ndarray:
    """Scatter compacted hypothesis records (sample_slot, orientation fields) into a fixed-slot
    table of pad_samples * R slots (uint8 view, pad_samples >= n_local_samples)."""
    tab = np.zeros(pad_samples * R, dtype=records.dtype)
    if len(records):
        slot = (records["sample_slot"].astype(np.int64) - slot_base) * R + records["orientation"]
        assert slot.min() >= 0 and slot.max() < n_local_samples * R
        tab[slot] = records
    return tab


def compact_table(tab: np.ndarray) -> np.ndarray:
    """Slot order IS output order: keep the occupied slots."""
    return tab[tab["n_points"] > 0]


def all_gather_tables(local_tab_u8, world: int, out=None):
    """all_gather_into_tensor of equal-sized uint8 tables (torch tensors on the backend's device).  `out`: a
    preallocated result tensor of world x the table's size (a caller that exchanges every step keeps one)."""
    import torch
    import torch.distributed as dist
    if out is None or out.numel() != local_tab_u8.numel() * world or out.device != local_tab_u8.device:
        out = torch.empty(local_tab_u8.numel() * world, dtype=torch.uint8, device=local_tab_u8.device)
    dist.all_gather_into_tensor(out, local_tab_u8)
    return out


# ---- spatial tiles: for clouds too large to replicate on every GPU ------------------------------
#
# A sample's hypotheses depend on the cloud only through the points within
# max(nn_radius_hands, nn_radius_taubin) of it and through those points' normals, which in turn
# depend on the points within normals_radius of each of them.  So a rank that owns the samples of
# one x-interval needs the points of that interval widened by the sum of the two radii and nothing
# else.  Three things make the result IDENTICAL to the unsplit run instead of merely close:
#   * the sample list is put in x order ONCE, independently of the number of ranks, and the RNG
#     stays keyed by the position in that list (slot_base + i);
#   * the tile keeps its points in their original relative order, so every neighbourhood is visited
#     in the same canonical (cell, index) order;
#   * the rank bins against the WHOLE cloud's minimum (ag2_set_grid_origin), so cells -- and the
#     cloud minimum z the prune test uses -- are those of the unsplit run.


def cloud_origin(xyz: np.ndarray) -> np.ndarray:
    """Per-axis minimum over the finite points, as float32: the unsplit run's grid origin."""
    a = np.asarray(xyz, dtype=np.float32)
    ok = np.isfinite(a).all(axis=1)
    return a[ok].min(axis=0).astype(np.float32)


def longest_axis(xyz: np.ndarray) -> int:
    """Axis along which tiles are thinnest relative to their halo: the cloud's longest extent."""
    a = np.asarray(xyz, dtype=np.float32)
    ok = np.isfinite(a).all(axis=1)
    return int(np.argmax(a[ok].max(axis=0) - a[ok].min(axis=0)))


def order_samples_by_x(xyz: np.ndarray, sample_idx: np.ndarray, axis: int = 0) -> np.ndarray:
    """The sample list in ascending x -- or another `axis` -- (stable): the rank-count-independent
    order tiles shard."""
    sample_idx = np.asarray(sample_idx)
    return sample_idx[np.argsort(np.asarray(xyz)[sample_idx, axis], kind="stable")]


def tile_halo(nn_radius_hands: float, nn_radius_taubin: float, normals_radius: float) -> float:
    """Width of the halo a tile needs either side of its samples' x-interval.  The last term covers
    the rounding of the f32 coordinate differences the radius tests see."""
    return max(nn_radius_hands, nn_radius_taubin) + normals_radius + 1e-5


def sample_costs(xyz: np.ndarray, ordered_sample_idx: np.ndarray, radius: float, axis: int = 0) -> np.ndarray:
    """Proxy of a sample's sweep cost: the number of cloud points whose coordinate along `axis` lies within
    `radius` of the sample's -- the 1-D marginal of K2, the neighbour count that drives the hand sweep
    (balance tiles on sum K2, not on the sample count).  Same on every rank."""
    x = np.asarray(xyz)[:, axis].astype(np.float64)
    xs = np.sort(x[np.isfinite(x)])
    sx = x[np.asarray(ordered_sample_idx, dtype=np.int64)]
    return (np.searchsorted(xs, sx + radius) - np.searchsorted(xs, sx - radius)).astype(np.float64)


def balanced_bounds(costs: np.ndarray, world: int) -> np.ndarray:
    """world + 1 boundaries of contiguous ranges of the ordered sample list with (nearly) equal summed
    cost; every rank gets at least one sample while there are enough."""
    n = len(costs)
    if n == 0:
        return np.zeros(world + 1, dtype=np.int64)
    if n <= world:                      # fewer samples than ranks: one each, the last ranks idle
        return np.minimum(np.arange(world + 1), n).astype(np.int64)
    cum = np.cumsum(np.maximum(np.asarray(costs, dtype=np.float64), 1.0))
    targets = cum[-1] * np.arange(1, world) / world
    b = np.concatenate([[0], np.searchsorted(cum, targets, side="left") + 1, [n]]).astype(np.int64)
    for r in range(1, world):           # at least one sample per rank
        b[r] = min(max(b[r], b[r - 1] + 1), n - (world - r))
    return b


def tile_points(xyz: np.ndarray, ordered_sample_idx: np.ndarray, rank: int, world: int,
                halo: float, axis: int = 0, bounds: np.ndarray | None = None):
    """Tile of `rank` along `axis` (the one the samples were ordered by): (keep, local_sample_idx,
    slot_base).

    keep              ascending indices of the points the rank must hold (apply the same subset to
                      the camera-source and normals arrays);
    local_sample_idx  the rank's samples as indices into xyz[keep];
    slot_base         position of the rank's first sample in ordered_sample_idx (RNG key and table
                      slot, as with replicated clouds)."""
    xyz = np.asarray(xyz)
    if bounds is None:
        b, e = shard_range(len(ordered_sample_idx), rank, world)
    else:               # cost-balanced ranges (balanced_bounds): same contract, other cut points
        b, e = int(bounds[rank]), int(bounds[rank + 1])
    mine = np.asarray(ordered_sample_idx[b:e], dtype=np.int64)
    if len(mine) == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int32), b
    x = xyz[:, axis].astype(np.float64)
    lo, hi = x[mine].min() - halo, x[mine].max() + halo
    keep = np.flatnonzero((x >= lo) & (x <= hi))      # NaN x compares false: dropped, never a neighbour
    local = np.searchsorted(keep, mine).astype(np.int32)
    assert (keep[local] == mine).all()
    return keep, local, b


# ---- compact exchange: only the occupied slots travel -----------------------------------------
#
# The fixed-slot table is mostly empty (cfg2: 2 400 of 40 000 slots), so a rank puts a 16-byte header
# {count, cap, 0, 0} and its min(count, cap) records in slot order on the wire
# (ag2_export_candidates_compact_device).  Records carry (sample_slot, orientation): the rank-order
# concatenation is the reference's output order, exactly compact_table() of the gathered full tables.

COMPACT_HEADER = 16


def compact_bytes(cap_records: int) -> int:
    return COMPACT_HEADER + cap_records * SLOT_BYTES


def pack_compact(records: np.ndarray, cap_records: int) -> np.ndarray:
    """Host twin of ag2_export_candidates_compact_device (records already in slot order)."""
    buf = np.zeros(compact_bytes(cap_records), dtype=np.uint8)
    buf[:8].view(np.uint32)[:] = (len(records), cap_records)
    n = min(len(records), cap_records)
    if n:
        buf[COMPACT_HEADER:COMPACT_HEADER + n * SLOT_BYTES] = records[:n].view(np.uint8).reshape(-1)
    return buf


def unpack_compact(gathered_u8: np.ndarray, world: int, cap_records: int, dtype):
    """(records of all ranks in rank order, truncated?) from the gathered compact buffers."""
    per = compact_bytes(cap_records)
    g = np.asarray(gathered_u8, dtype=np.uint8).reshape(world, per)
    parts, cut = [], False
    for r in range(world):
        count, cap = (int(v) for v in g[r, :8].view(np.uint32))
        assert cap == cap_records
        cut = cut or count > cap
        n = min(count, cap)
        parts.append(g[r, COMPACT_HEADER:COMPACT_HEADER + n * SLOT_BYTES].copy().view(dtype))
    return (np.concatenate(parts) if parts else np.zeros(0, dtype)), cut
