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


def all_gather_tables(local_tab_u8, world: int):
    """all_gather_into_tensor of equal-sized uint8 tables (torch tensors on the backend's device)."""
    import torch
    import torch.distributed as dist
    out = torch.empty(local_tab_u8.numel() * world, dtype=torch.uint8, device=local_tab_u8.device)
    dist.all_gather_into_tensor(out, local_tab_u8)
    return out
