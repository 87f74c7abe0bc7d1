"""N > 1 path on CPU: world_size-2 gloo processes shard the sample list by range (cloud replicated,
or cut into spatial tiles as bench.py does), each fills its
fixed-slot candidate table, one all-gather exchanges them, and the merged result must equal the
single-process result record for record -- whatever the number of ranks.  The per-rank compute is
done by the oracle here (no GPU in this container); the sharding, slot addressing, global-slot RNG
keying and the collective are exactly the code bench.py runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q, tiles=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from agile_grasp2_amd import scene, sharding
    from conftest import scene_params
    from oracle import api
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    xyz, ws = scene.make_scene(seed=3, n_target=6000)
    idx = scene.draw_samples(3, xyz.shape[0], 61)   # odd count: uneven shards
    R = 8
    o = api.Oracle(**scene_params(ws, num_threads=2))
    b, e = sharding.shard_range(len(idx), rank, world)
    if tiles:
        # spatial tiles (what bench.py does for N > 1): the rank holds only its x-interval + halo
        ordered = sharding.order_samples_by_x(xyz, idx)
        q_ = o.params
        halo = sharding.tile_halo(q_.nn_radius_hands, q_.nn_radius_taubin, q_.normals_radius)
        keep, local, base = sharding.tile_points(xyz, ordered, rank, world, halo)
        assert base == b and len(local) == e - b and len(keep) <= xyz.shape[0]
        o.set_grid_origin(sharding.cloud_origin(xyz))
        o.set_cloud(xyz[keep])
        o.compute_normals()
        recs = o.generate_hypotheses(sample_idx=local, slot_base=b, seed=11)
    else:
        o.set_cloud(xyz)
        o.compute_normals()
        recs = o.generate_hypotheses(sample_idx=idx[b:e], slot_base=b, seed=11)
    pad = sharding.max_shard(len(idx), world)
    tab = sharding.table_from_records(recs, b, e - b, R, pad)
    local = torch.from_numpy(tab.view(np.uint8).copy())
    gathered = sharding.all_gather_tables(local, world).numpy().view(api.HYP_DTYPE)
    merged = sharding.compact_table(gathered)
    # the compact exchange (only occupied slots on the wire) must deliver the same records
    cap = 64
    while True:
        local_c = torch.from_numpy(sharding.pack_compact(sharding.compact_table(tab), cap))
        got, cut = sharding.unpack_compact(sharding.all_gather_tables(local_c, world).numpy(), world, cap,
                                           api.HYP_DTYPE)
        if not cut:
            break
        cap *= 4        # a rank's list was cut: every rank sees it in the headers and retries
    assert got.tobytes() == merged.tobytes()
    if rank == 0:
        q.put(merged.tobytes())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,tiles", [(2, False), (3, False), (2, True), (3, True)])
def test_sharded_all_gather_equals_single_process(world, tiles):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from agile_grasp2_amd import scene, sharding
    from conftest import scene_params
    from oracle import api
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, tiles)) for r in range(world)]
    for p in procs:
        p.start()
    blob = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    merged = np.frombuffer(blob, dtype=api.HYP_DTYPE)
    xyz, ws = scene.make_scene(seed=3, n_target=6000)
    idx = scene.draw_samples(3, xyz.shape[0], 61)
    o = api.Oracle(**scene_params(ws, num_threads=2))
    o.set_cloud(xyz)
    o.compute_normals()
    if tiles:  # the tiled job works through the samples in x order
        idx = sharding.order_samples_by_x(xyz, idx)
    want = o.generate_hypotheses(sample_idx=idx, slot_base=0, seed=11)
    assert len(want) > 10
    assert merged.tobytes() == want.tobytes()


def test_shard_ranges_cover_and_balance():
    from agile_grasp2_amd import sharding
    for n in (0, 1, 7, 61, 5000):
        for w in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) <= sharding.max_shard(n, w)
