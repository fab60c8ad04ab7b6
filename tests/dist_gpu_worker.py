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
    if case.get("mode") == "supervised":
        return supervised(rank, world, port, outdir, case)
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


def supervised(rank, world, port, outdir, case):
    """The reference's supervised loop body (src/supervised.py:863-868, 897-902) on this rank's shard; the process group
    is the one main() initialised (ddp_setup is idempotent)."""
    import jolineedle_amd as ja
    from jolineedle_amd.dist import shard_range
    from tests.helpers import make_pair
    P, T, B = case["P"], case["T"], case["B"]
    product, _ = make_pair(5, patch_size=P, block_size=T, with_detector=False, image_processor=None, max_batch=B * T)
    lo, hi = shard_range(B, rank, world)
    batch = {"image": case["images"][lo:hi].cuda(), "bboxes": case["bboxes"][lo:hi]}
    cfg = ja.CfgNode(patch_size=P, max_seq_len=T, min_keypoints=0, max_keypoints=2, binomial_keypoints=False, stop_enabled=True,
                     stop_weight=1.0, learning_rate=1e-3, gradient_accumulation=1, detection_enabled=False, max_iters=2)
    trainer = ja.SupervisedTrainer(cfg, product, rank=rank)
    trainer.ddp_setup(rank, world, port, backend="gloo")
    optim_gpt, optim_yolox = product.configure_optimizers(cfg)
    assert optim_yolox is None and optim_gpt.sync_gradients
    product.train()
    tr = trainer.generate_trajectories(batch, seed=100 + 10 * rank)
    action_logits, _ = product(tr["patches"], tr["current_actions"], classes=tr["class_id"], positions=tr["positions"])
    metrics = trainer.compute_metrics(action_logits, tr["next_actions"], tr["masks"])
    metrics["loss"].backward()
    named = dict(product.named_parameters())
    local = {k: p.grad.detach().cpu().clone() for k, p in named.items() if p.grad is not None}
    optim_gpt.step()                     # ONE all-reduce of the flat buffer (SUM) + AdamW on the mean; no clipping
    mean = {k: p.grad.detach().cpu().clone() / world for k, p in named.items() if p.grad is not None}
    optim_gpt.zero_grad()
    params = {k: v.detach().cpu().clone() for k, v in product.named_parameters()}
    world_seen = dist.get_world_size()
    # ... and run() itself for two more iterations on the same group (main() created it, so run() must leave it alone)
    m = trainer.run(rank, world, port, batches=[batch], max_iters=2, backend="gloo", seed=7 + rank)
    assert dist.is_initialized(), "run() destroyed a process group it did not create"
    torch.save({"local": local, "mean": mean, "world_seen": world_seen, "params": params,
                "trajectories": {k: v.detach().cpu() for k, v in tr.items() if isinstance(v, torch.Tensor)},
                "run_iters": trainer.iter_num, "run_loss_finite": bool(torch.isfinite(m["loss"]))}, outdir / f"rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
