"""World-size-2 gloo tests (CPU) of the N > 1 path: batch sharding and the single flat
gradient all-reduce + mean scaling."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jolineedle_amd.dist import allreduce_gradients, shard_range


def test_shard_range_partitions_the_batch():
    for gb, world in ((512, 8), (64, 1), (10, 4), (3, 8), (0, 2)):
        spans = [shard_range(gb, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == gb
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [e - b for b, e in spans]
        assert max(sizes) - min(sizes) <= 1
    assert shard_range(512, 3, 8) == (192, 256)        # BASELINE config 4: 64 agents per GPU


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    n_total, n_opt = 1000, 800                           # arena: [optim_gpt | yolox]
    flat = torch.randn(n_total)
    mine = flat.clone()
    timing = []
    scale = allreduce_gradients(flat, n_opt, timing=timing)
    assert timing == []                                  # event pairs are recorded for device tensors only (bench.py --gpus N)
    gathered = [torch.zeros(n_total) for _ in range(world)]
    dist.all_gather(gathered, mine)
    want = sum(g[:n_opt] for g in gathered)
    ok = torch.allclose(flat[:n_opt], want) and torch.equal(flat[n_opt:], mine[n_opt:]) and scale == 1.0 / world
    # ranks that apply the same optimiser step to the mean gradient stay bit-identical
    p = torch.ones(n_opt) - 0.1 * (flat[:n_opt] * scale)
    ps = [torch.zeros(n_opt) for _ in range(world)]
    dist.all_gather(ps, p)
    ok = ok and all(torch.equal(ps[0], q) for q in ps)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def test_allreduce_is_identity_without_process_group():
    flat = torch.arange(10.0)
    assert allreduce_gradients(flat, 5) == 1.0 and torch.equal(flat, torch.arange(10.0))
