"""Data-parallel glue: the rollout path shards by image (one rank per GPU, independent agents,
no collective inside a trajectory); the only exchange step of a REINFORCE iteration is ONE
all-reduce of the flat gradient arena (RCCL over xGMI with backend "nccl"; gloo on CPU in tests).

The reference has no gradient sync in RL mode (src/reinforce.py:279-280, DDP commented out) and
DDP mean-reduction in supervised mode (src/supervised.py:815): ranks average their gradients.
"""
from typing import Tuple

import torch


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of the images rank `rank` owns: contiguous blocks, sizes differ by at most one
    (DistributedSampler-style partition, src/reinforce.py:284-294, without padding duplicates)."""
    assert 0 <= rank < world and global_batch >= 0
    base, rem = divmod(global_batch, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def allreduce_gradients(flat: torch.Tensor, numel: int, group=None) -> float:
    """SUM all-reduce of flat[:numel] in place; returns the scale (1 / world_size) the optimiser
    applies so that ranks step with the MEAN gradient."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat[:numel], op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world
