"""Rank process of tests/test_gpu_parity.py::test_two_ranks_share_one_gpu_and_average_gradients: two of these share cuda:0
over gloo (the driver's multi-GPU node is not ours to use; the data path is the same — one rank per process, independent
agents per rank, ONE all-reduce of the flat gradient arena per optimiser step).
usage: dist_gpu_worker.py RANK WORLD PORT OUTDIR"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch
import torch.distributed as dist


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), Path(sys.argv[4])
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import jolineedle_amd as ja
    from jolineedle_amd import _lib
    from jolineedle_amd.dist import allreduce_gradients, shard_range
    from tests.helpers import make_pair, synth_batch
    case = torch.load(outdir / "case.pt")
    P, Tn, B = case["P"], case["T"], case["B"]
    product, _ = make_pair(5, patch_size=P, block_size=Tn, with_detector=False, image_processor=None)
    lo, hi = shard_range(B, rank, world)
    images, bboxes, start, forced = (case[k][lo:hi] for k in ("images", "bboxes", "start", "forced"))
    cfg = ja.CfgNode(max_seq_len=Tn, entropy_weight=0.01, stop_enabled=True, reward_norm=True, seed=0,
                     learning_rate=1e-3, gradient_accumulation=1)
    tr = ja.ReinforceTrainer(cfg, product)
    tr.last_return_mean, tr.last_return_std = 0.25, 1.5
    env = ja.NeedleGeneralEnv(images.cuda(), bboxes, P, Tn, 1, True)
    tr.train_iteration(env, forced_actions=forced, start_positions=start, optimizer_step=False)
    local = product.engine_grads()
    arena = product.grad_arena()
    scale = allreduce_gradients(arena, product._optim_gpt_numel)
    mean = {k: v * scale for k, v in product.engine_grads().items()}
    eng = product.engine()
    _lib.check(eng.lib.jn_optimizer_step(eng.handle, 1e-3, 0.01, 1.0, scale, _lib.current_stream(product.device)), "step")
    product.pull_parameters()
    torch.save({"local": local, "mean": mean, "world_seen": dist.get_world_size(),
                "params": {k: v.detach().cpu().clone() for k, v in product.named_parameters()}}, outdir / f"rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
