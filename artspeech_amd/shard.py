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
