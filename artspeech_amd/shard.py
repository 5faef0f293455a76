"""Batch sharding across GPUs: utterances are independent, so a global batch is split across ranks with NO
data-path collective.  Length-sorted round-robin keeps the per-rank frame counts balanced;
results come back in the caller's order.  torch.distributed is used only to gather results / timings."""
from typing import List, Sequence


def shard_indices(lengths: Sequence[int], world_size: int, rank: int) -> List[int]:
    """Indices of the utterances rank `rank` processes: sort by length (desc, stable), deal round-robin."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return sorted(order[rank::world_size])


def all_shards(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    return [shard_indices(lengths, world_size, r) for r in range(world_size)]


def gather_objects(local, world_size: int):
    """all-gather python objects (host side; e.g. per-utterance mels as numpy arrays) -- not on the data path."""
    if world_size == 1:
        return [local]
    import torch.distributed as dist
    out = [None] * world_size
    dist.all_gather_object(out, local)
    return out


def merge(shards: List[List[int]], results: List[list], n: int) -> list:
    """inverse of sharding: results[r][k] belongs to utterance shards[r][k]."""
    merged = [None] * n
    for idx, res in zip(shards, results):
        for i, v in zip(idx, res):
            merged[i] = v
    return merged


def split_utterances(packed, lens):
    """packed [C][sum lens] (packed frames, numpy or tensor) -> list of per-utterance [C][len] arrays"""
    out, o = [], 0
    for n in lens:
        out.append(packed[:, o:o + int(n)])
        o += int(n)
    return out


def merge_shards(gathered, n: int) -> list:
    """gathered: per rank (indices, per-utterance results) -> the results in the caller's order"""
    merged = [None] * n
    for idx, res in gathered:
        for i, v in zip(idx, res):
            merged[i] = v
    if any(v is None for v in merged):
        raise ValueError("merge_shards: some utterances were not produced by any rank")
    return merged


def sharded_forward(step_fn, lengths: Sequence[int], world_size: int = 1, rank: int = 0, gather=None):
    """The C4 path (BASELINE: one global batch sharded over the GPUs of a node): this rank runs `step_fn(indices)` -> the list of
    per-utterance results of ITS utterances (length-sorted round-robin shard, no data-path collective); rank 0 gets every
    utterance's result back in the caller's order (other ranks: None).  gather: callable(local) -> list over ranks on rank 0
    (default: torch.distributed.gather_object); with world_size 1 nothing is communicated."""
    mine = shard_indices(lengths, world_size, rank)
    local = (mine, list(step_fn(mine)))
    if len(local[1]) != len(mine):
        raise ValueError("sharded_forward: step_fn must return one result per index")
    if world_size == 1:
        return merge_shards([local], len(lengths))
    if gather is None:
        import torch.distributed as dist

        def gather(obj):
            out = [None] * world_size if rank == 0 else None
            dist.gather_object(obj, out, dst=0)
            return out
    gathered = gather(local)
    return merge_shards(gathered, len(lengths)) if rank == 0 else None
