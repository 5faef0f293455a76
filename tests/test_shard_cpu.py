"""CPU, world_size 2 over gloo: the multi-GPU path's host logic (sharding, no data-path collective, result
merge, max-over-ranks timing) with the oracle standing in for the device computation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from artspeech_amd import shard


def test_shard_indices_cover_and_balance():
    lengths = [40, 12, 30, 5, 40, 23, 8, 17, 33]
    shards = shard.all_shards(lengths, 2)
    assert sorted(shards[0] + shards[1]) == list(range(len(lengths)))
    loads = [sum(lengths[i] for i in s) for s in shards]
    assert abs(loads[0] - loads[1]) <= max(lengths)
    assert shard.shard_indices(lengths, 1, 0) == list(range(len(lengths)))
    with pytest.raises(ValueError):
        shard.shard_indices(lengths, 2, 2)
    assert shard.all_shards([], 2) == [[], []]
    # packed results back to utterances, and the merge of (indices, results) pairs
    packed = np.arange(2 * 9).reshape(2, 9)
    parts = shard.split_utterances(packed, [4, 0, 5])
    assert [p.shape for p in parts] == [(2, 4), (2, 0), (2, 5)] and np.array_equal(parts[2], packed[:, 4:])
    assert shard.merge_shards([([2, 0], ["c", "a"]), ([1], ["b"])], 3) == ["a", "b", "c"]
    with pytest.raises(ValueError):
        shard.merge_shards([([0], ["a"])], 2)
    assert shard.sharded_forward(lambda idx: [10 * i for i in idx], [3, 1, 2]) == [0, 10, 20]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle import mas as omas
    rng = np.random.default_rng(0)
    n = 7
    values = [rng.random((1, 6 + i, 20 + 3 * i), dtype=np.float32) for i in range(n)]
    lengths = [v.shape[2] for v in values]
    mine = shard.shard_indices(lengths, world, rank)
    local = [omas.maximum_path_c(values[i], np.ones_like(values[i]), False) for i in mine]
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = shard.gather_objects(local, world)
    merged = shard.merge(shard.all_shards(lengths, world), gathered, n)
    ok = all(np.array_equal(merged[i], omas.maximum_path_c(values[i], np.ones_like(values[i]), False)) for i in range(n))
    # the product entry point of the C4 path (bench.py --global-batch, pipeline.synthesis_mel(world=...)): same code, the device step
    # stubbed by a function of the utterance (here: the oracle's MAS; on the GPU: ArtsSpeech.forward_packed on the shard)
    calls = []

    def step(indices):
        calls.append(list(indices))
        return [omas.maximum_path_c(values[i], np.ones_like(values[i]), False) for i in indices]

    merged2 = shard.sharded_forward(step, lengths, world, rank)
    assert calls == [mine]
    if rank == 0:
        ok = ok and all(np.array_equal(merged2[i], merged[i]) for i in range(n))
    else:
        ok = ok and merged2 is None
    if rank == 0:
        q.put((ok, float(t.item())))
    dist.destroy_process_group()


def test_two_rank_gloo_roundtrip():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and abs(tmax - 0.2) < 1e-12
