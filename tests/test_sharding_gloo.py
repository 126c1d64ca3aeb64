"""The N > 1 path (keyframe sharding + result gather) on CPU: world_size 2 and 3, gloo backend."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simplerecon_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        idx = sharding.shard_indices(n_items, rank, world)
        # the "depth map" of keyframe i is a 2x3 tensor filled with i (what a replica would compute)
        local = torch.stack([torch.full((1, 2, 3), float(i)) for i in idx]) if idx else torch.zeros((0, 1, 2, 3))
        out = sharding.gather_results(local, n_items, dst=0)
        if rank == 0:
            ok = out.shape == (n_items, 1, 2, 3) and all(bool((out[i] == i).all()) for i in range(n_items))
            q.put(bool(ok))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 8), (2, 7), (3, 10), (2, 1)])
def test_round_robin_shard_and_gather(world, n_items):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() is True


def test_shard_indices_partition():
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 2048):
            seen = sorted(i for r in range(world) for i in sharding.shard_indices(n, r, world))
            assert seen == list(range(n))
    assert sharding.shard_indices(2048, 3, 8)[:3] == [3, 11, 19]
    assert len(sharding.shard_indices(2048, 0, 8)) == 256  # BASELINE.json configs[3]: 256 keyframes per GPU
    assert sharding.batches(list(range(10)), 4) == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]
    with pytest.raises(ValueError):
        sharding.shard_indices(4, 2, 2)


def test_bench_stream_batches_partition_the_stream():
    """bench.py --workload hero_cfg4_stream: every keyframe of the world x steps x 8 stream is processed exactly once,
    by rank id mod world (BASELINE.json configs[3]: 2048 keyframes over 8 GPUs = 32 batches of 8 per rank)."""
    import bench_workloads as bw
    world, batch, steps = 8, 8, 32
    seen = []
    for rank in range(world):
        for step in range(steps):
            ids = bw.stream_batch_ids(rank, world, batch, steps, step)
            assert len(ids) == batch and all(i % world == rank for i in ids)
            seen += ids
    assert sorted(seen) == list(range(2048))
