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


def allreduce_gradients(flat: torch.Tensor, numel: int, group=None, timing: list = None) -> float:
    """SUM all-reduce of flat[:numel] in place; returns the scale (1 / world_size) the optimiser
    applies so that ranks step with the MEAN gradient.

    Stream order: the call is issued on torch's CURRENT stream — the stream the engine's backward was
    launched on.  With backend "nccl" (RCCL) ProcessGroupNCCL runs the collective on its own stream, which first waits for
    an event recorded on the current stream at call time (the whole backward: its last kernel, the stem's weight gradient,
    is what completes the arena) and which the current stream then waits for (synchronous op): the optimiser kernel that
    follows on the current stream sees the reduced gradients, and the host is never blocked.  `timing`: a list that
    receives one (start, end) pair of CUDA events recorded on the current stream around the collective (bench.py's
    `allreduce_ms`; the first pair of a process includes RCCL's lazy communicator set-up)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world > 1:
        ev = None
        if timing is not None and flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        dist.all_reduce(flat[:numel], op=dist.ReduceOp.SUM, group=group)
        if ev is not None:
            ev[1].record()
            timing.append(ev)
    return 1.0 / world
