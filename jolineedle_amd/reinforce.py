"""``ReinforceTrainer.rollout`` and the REINFORCE metrics of the reference
(src/reinforce.py:73-265) over ``jn_rollout``: the whole trajectory of the whole batch is
enqueued on the current stream, with no host synchronisation until the step count is read."""
import ctypes as C
from typing import Dict

import torch

from . import _lib
from ._lib import JnRolloutOut, JnTrainOpts, check, ptr
from .env import NeedleGeneralEnv


class _RolloutGraph(torch.autograd.Function):
    """Graph node behind ``rollout()["logprobs"]`` / ``["entropies"]`` of a train-mode rollout ("carry
    autograd graph in training").  backward = ``jn_reinforce_backward`` with the upstream gradients torch hands over, then
    the engine's packed gradients are added to ``param.grad`` (reference layout)."""

    @staticmethod
    def forward(ctx, anchor, model, gen, logprobs, entropies, keep):
        # `keep`: every buffer the engine's backward reads through raw pointers (logits, actions, positions, final_emb,
        # masks, ..., forced actions) — held here so that the caching allocator cannot hand them out again when the
        # caller drops the rollout dict before loss.backward()
        ctx.model, ctx.gen, ctx.keep = model, gen, keep
        return logprobs.view_as(logprobs), entropies.view_as(entropies)

    @staticmethod
    def backward(ctx, dlp, dent):
        model = ctx.model
        if ctx.gen != model._rollout_gen:
            raise RuntimeError("backward through a rollout whose activations were overwritten by a later train-mode rollout")
        eng, dev = model.engine(), model.device
        dlp = None if dlp is None else dlp.to(dev, torch.float32).contiguous()
        dent = None if dent is None else dent.to(dev, torch.float32).contiguous()
        check(eng.lib.jn_reinforce_backward(eng.handle, ptr(dlp), ptr(dent), _lib.current_stream(dev)), "jn_reinforce_backward")
        model.publish_engine_grads()
        return None, None, None, None, None, None


class ReinforceTrainer:
    """Rollout / loss part of the reference trainer (dataset, Visdom, checkpoints are
    out of scope).  ``config`` needs max_seq_len, entropy_weight,
    stop_enabled, reward_norm (main.py:310-388)."""

    def __init__(self, config, model, logger=None, train_dataset=None, test_dataset=None, rank: int = 0):
        self.config, self.model, self.logger, self.rank = config, model, logger, rank
        self.patch_size = model.patch_size
        self.max_ep_len = config.max_seq_len
        self.entropy_weight = config.entropy_weight
        self.n_glimps_levels = 1
        self.stop_enabled = config.stop_enabled
        self.device = model.device
        self.best_metric_name = "prop_patches_found"
        self.last_return_values = []
        self.last_return_mean = 0
        self.last_return_std = 1
        self.seed = int(getattr(config, "seed", 0))
        self._rollouts = 0

    def yolox_model(self):
        return self.model.yolox

    @torch.no_grad()
    def _compute_last_returns_mean_std(self):
        allv = torch.cat([v.float().cpu() for v in self.last_return_values]) if self.last_return_values else torch.zeros(0)
        if len(allv) == 0:
            mean, std = 0, 1
        elif len(allv) == 1:
            mean, std = allv[0], 1
        else:
            mean, std = allv.mean(), allv.std()
        self.last_return_mean, self.last_return_std = mean, std
        self.last_return_values = []

    def rollout(self, env: NeedleGeneralEnv, do_detection: bool = False, sample_actions: bool = True,
                forced_actions: torch.Tensor = None, start_positions: torch.Tensor = None,
                keep_patches: bool = True, stop_early: bool = True) -> Dict[str, torch.Tensor]:
        """src/reinforce.py:108-215.  Extra keyword arguments (not in the reference):
        `forced_actions` [B,T] replays a trajectory, `start_positions` [B,2] injects reset
        positions (reference: env.reset(positions)), `keep_patches=False` skips the
        [B,S+1,3,P,P] patch stack, `stop_early=False` always runs max_ep_len steps."""
        model, dev = self.model, self.device
        model.sync_weights()
        eng = model.engine()
        env.bind(eng)
        B, T, P = env.batch_size, env.max_ep_len, env.patch_size
        C_, nA = model.n_embd, eng.cfg.n_actions
        f32 = dict(device=dev, dtype=torch.float32)
        buf = {
            "rewards": torch.empty((B, T), **f32), "returns": torch.empty((B, T), **f32),
            "logprobs": torch.empty((B, T), **f32), "entropies": torch.empty((B, T), **f32),
            "masks": torch.empty((B, T + 1), device=dev, dtype=torch.uint8),
            "logit_masks": torch.empty((B, T), device=dev, dtype=torch.uint8),
            "positions": torch.empty((B, T + 1, 2), device=dev, dtype=torch.int64),
            "actions": torch.empty((B, T), device=dev, dtype=torch.int64),
            "logits": torch.empty((B, T, nA), **f32),
            "final_emb": torch.empty((B, T + 1, C_), **f32),
        }
        patches = torch.empty((B, T + 1, 3, P, P), **f32) if keep_patches else None
        out = JnRolloutOut()
        for k, t in buf.items():
            setattr(out, k + "_dev", t.data_ptr())
        out.patches_dev = patches.data_ptr() if patches is not None else None
        Kd = eng.cfg.max_det_per_patch
        if do_detection:
            det_boxes = torch.zeros((B, T + 1, Kd, 7), **f32)
            det_counts = torch.zeros((B, T + 1), device=dev, dtype=torch.int32)
            out.det_boxes_dev, out.det_counts_dev = det_boxes.data_ptr(), det_counts.data_ptr()
        if forced_actions is not None:
            mode = _lib.JN_MODE_FORCED
            forced_actions = forced_actions.to(dev, torch.int64).contiguous()
        else:
            mode = _lib.JN_MODE_SAMPLE if sample_actions else _lib.JN_MODE_GREEDY
        if start_positions is not None:
            start_positions = start_positions.to(dev, torch.int64).contiguous()
        self._rollouts += 1
        seed = (self.seed * 1000003 + self._rollouts) & 0xFFFFFFFFFFFFFFFF
        stream = _lib.current_stream(dev)
        # model.train() + grad mode (the reference's training loop, src/reinforce.py:304, 326): batch-statistics BatchNorm,
        # activations of every step kept resident, logprobs / entropies leave with a graph (autograd bridge)
        graph = bool(model.training) and torch.is_grad_enabled() and not do_detection
        if graph:
            model.bind_flat()
            model._rollout_gen += 1
            check(eng.lib.jn_reinforce_forward(eng.handle, mode, ptr(forced_actions), ptr(start_positions), seed,
                                               int(stop_early), C.byref(out), stream), "jn_reinforce_forward")
            anchor = next(p for p in model.parameters() if p.requires_grad)
            keep = dict(buf, forced_actions=forced_actions, start_positions=start_positions)
            buf["logprobs"], buf["entropies"] = _RolloutGraph.apply(anchor, model, model._rollout_gen, buf["logprobs"],
                                                                    buf["entropies"], keep)
        else:
            check(eng.lib.jn_rollout(eng.handle, mode, ptr(forced_actions), ptr(start_positions), seed,
                                     int(do_detection), int(stop_early), C.byref(out), stream), "jn_rollout")
        S = C.c_int()
        check(eng.lib.jn_rollout_steps(eng.handle, C.byref(S), stream), "jn_rollout_steps")
        S = S.value
        res = {
            "rewards": buf["rewards"][:, :S], "returns": buf["returns"][:, :S],
            "logprobs": buf["logprobs"][:, :S], "entropies": buf["entropies"][:, :S],
            "masks": buf["masks"][:, :S + 1].bool(), "logit_masks": buf["logit_masks"][:, :S].bool(),
            "positions": buf["positions"][:, :S + 1], "bboxes": [[] for _ in range(B)],
            "det_counts": det_counts[:, :S + 1] if do_detection else None,
            "patches": patches[:, :S + 1] if patches is not None else None,
            "actions": buf["actions"][:, :S], "logits": buf["logits"][:, :S],
            "final_emb": buf["final_emb"][:, :S + 1],
        }
        if do_detection:                      # ragged list-of-lists of [n,7] | None (src/reinforce.py:145-146, 166-167)
            cnt = det_counts[:, :S + 1].tolist()
            res["bboxes"] = [[det_boxes[b, t, :cnt[b][t]].clone() if cnt[b][t] > 0 else None for t in range(S + 1)]
                             for b in range(B)]
        return res

    # ---- the reference's training loop on the autograd bridge (src/trainer.py:61-71, src/reinforce.py:267-362) --------
    def ddp_setup(self, rank: int, world_size: int, port: int, backend: str = None):
        """``Trainer.ddp_setup`` (src/trainer.py:61-71): one process per GPU, backend "nccl" (= RCCL over xGMI on ROCm).
        Rendezvous on 127.0.0.1 (the reference's "localhost" may not resolve in a container); ``backend="gloo"`` is for
        CPU-side rehearsals."""
        import os
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        if not dist.is_initialized():
            dist.init_process_group(backend=backend or "nccl", rank=rank, world_size=world_size)
            self._owns_process_group = True           # run() only tears down a group it created itself

    def training_step(self, batch, optim_gpt, optim_yolox=None, sync_gradients: bool = True, **rollout_kw):
        """The body of the reference's loop, statement for statement (src/reinforce.py:302-353): env -> rollout ->
        compute_metrics -> [detector loss] -> (loss / ga).backward() -> every ga-th iteration clip_grad_value_ ->
        optim.step() -> optim.zero_grad() -> reward-norm window.  Everything torch sees is real: ``loss`` has a graph (the
        rollout node calls the engine's backward), ``param.grad`` are tensors, the optimisers are ``torch.optim.Optimizer``s."""
        from torch.nn.utils import clip_grad
        config, model = self.config, self.model
        self.iter_num = getattr(self, "iter_num", 0) + 1
        model.train()
        images, bboxes = batch["image"].to(self.device), batch["bboxes"]
        env = NeedleGeneralEnv(images, bboxes, self.patch_size, self.max_ep_len, self.n_glimps_levels, self.stop_enabled,
                               engine=model.engine())
        rollout = self.rollout(env, keep_patches=False, **rollout_kw)
        metrics = self.compute_metrics(rollout)
        loss = metrics["loss"]
        ga = int(getattr(config, "gradient_accumulation", 1))
        if getattr(config, "detection_enabled", False) and self.yolox_model() is not None:      # src/reinforce.py:330-339
            patches_yolox, bboxes_yolox = env.get_detection_batch(int(getattr(config, "detection_sample_neg", 1)))
            if getattr(self, "detection_augment", None) is not None:
                with torch.no_grad():
                    patches_yolox = self.detection_augment(patches_yolox)
            yolox = self.yolox_model()
            _, _, yolo_loss = yolox(patches_yolox, bboxes_yolox, predict=False)      # (the loop discards outputs / fpn_outs)
            for k, v in yolo_loss.items():
                metrics["yolo_" + k] = v.detach()
            total_loss = yolo_loss["total_loss"]         # carries the detector's graph (yolox.py::_DetectorGraph)
            loss = loss + total_loss
        (loss / ga).backward()                           # :341 — ONE backward through the rollout node and the detector node
        if self.iter_num % ga == 0:
            opts = [o for o in (optim_gpt, optim_yolox) if o is not None]
            was = [getattr(o, "sync_gradients", False) for o in opts]
            if sync_gradients:
                # data-parallel ranks: ONE all-reduce of the flat gradient buffer (mean), before the clip as under DDP —
                # the optimisers must not reduce the (already averaged, clipped) gradients a second time
                from .dist import allreduce_gradients
                sc = allreduce_gradients(model._flat_grads, model._arena_numel)
                if sc != 1.0:
                    model._flat_grads.mul_(sc)
                for o in opts:
                    if hasattr(o, "sync_gradients"):
                        o.sync_gradients = False
            clip_grad.clip_grad_value_(model.parameters(), 1)
            optim_gpt.step()
            optim_gpt.zero_grad()
            if getattr(config, "detection_enabled", False) and optim_yolox is not None:
                optim_yolox.step()
                optim_yolox.zero_grad()
            for o, w in zip(opts, was):
                if hasattr(o, "sync_gradients"):
                    o.sync_gradients = w
            if config.reward_norm:
                self._compute_last_returns_mean_std()
        metrics["loss"] = loss.detach()
        return metrics

    def run(self, rank: int, world_size: int, ddp_port: int, batches=None, max_iters: int = None, backend: str = None):
        """``ReinforceTrainer.run`` (src/reinforce.py:267-362) without the dataset / Visdom / test plumbing (out of scope,
        SURVEY.md §8): `batches` is any iterable of collated batches ({"image": [B,3,H,W], "bboxes": [B,nb,4]}, the
        ``padded_collate_fn`` layout; default: ``self.train_dataset``), re-iterated when exhausted.  Returns the metrics of
        the last iteration."""
        import torch.distributed as dist
        if getattr(self.config, "detection_enabled", False) and getattr(self.config, "augment_detection", False):
            self.init_detection()
        self.rank = rank
        if world_size > 1 or backend:
            self.ddp_setup(rank, world_size, ddp_port, backend)
        model = self.model
        optim_gpt, optim_yolox = model.configure_optimizers(self.config)
        for o in (optim_gpt, optim_yolox):
            if o is not None:
                o.sync_gradients = False                     # training_step averages the gradients before the clip
        self.optim_gpt, self.optim_yolox = optim_gpt, optim_yolox
        batches = batches if batches is not None else self.train_dataset
        assert batches is not None, "run() needs an iterable of collated batches"
        n_iters = int(max_iters if max_iters is not None else getattr(self.config, "max_iters", 1))
        it = iter(batches)
        metrics = None
        for _ in range(n_iters):
            try:
                batch = next(it)
            except StopIteration:
                it = iter(batches)
                batch = next(it)
            metrics = self.training_step(batch, optim_gpt, optim_yolox, sync_gradients=world_size > 1)
        if getattr(self, "_owns_process_group", False) and dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()                 # src/reinforce.py:362 / src/supervised.py:911
            self._owns_process_group = False
        return metrics

    def init_detection(self, **kw):
        """``Trainer.init_detection`` (src/trainer.py:176-186): installs the on-device augmentation of the detector
        patches.  Off until called (the parity tests compare un-augmented steps); keyword arguments go to
        ``augment.DetectionAugment`` (e.g. ``planckian_coeffs``)."""
        from .augment import DetectionAugment
        kw.setdefault("seed", int(getattr(self.config, "seed", 0)) + 17 * int(getattr(self, "rank", 0)))
        self.detection_augment = DetectionAugment(**kw)
        return self.detection_augment

    # ---- training: one REINFORCE iteration (src/reinforce.py:302-353) ------------------------
    def _grad_arena(self):
        """The engine's flat fp32 gradient arena: owned by the MODEL (one per engine, shared by all trainers)."""
        g = self.model.grad_arena()
        self._optim_numel = self.model._optim_gpt_numel
        return g

    def train_iteration(self, env: NeedleGeneralEnv, sample_actions: bool = True, forced_actions=None,
                        start_positions=None, stop_early: bool = True, optimizer_step: bool = True,
                        process_group=None) -> Dict[str, torch.Tensor]:
        """rollout (train-mode BatchNorm) -> compute_metrics -> backward -> [all-reduce] -> clip + AdamW,
        all inside the engine.  Returns the metrics of src/reinforce.py:243-252."""
        model, dev = self.model, self.device
        model.sync_weights()
        eng = model.engine()
        env.bind(eng)
        grads = self._grad_arena()
        B, T, P = env.batch_size, env.max_ep_len, env.patch_size
        C_, nA = model.n_embd, eng.cfg.n_actions
        f32 = dict(device=dev, dtype=torch.float32)
        buf = {
            "rewards": torch.empty((B, T), **f32), "returns": torch.empty((B, T), **f32),
            "logprobs": torch.empty((B, T), **f32), "entropies": torch.empty((B, T), **f32),
            "masks": torch.empty((B, T + 1), device=dev, dtype=torch.uint8),
            "logit_masks": torch.empty((B, T), device=dev, dtype=torch.uint8),
            "positions": torch.empty((B, T + 1, 2), device=dev, dtype=torch.int64),
            "actions": torch.empty((B, T), device=dev, dtype=torch.int64),
            "logits": torch.empty((B, T, nA), **f32),
            "final_emb": torch.empty((B, T + 1, C_), **f32),
        }
        out = JnRolloutOut()
        for k, t in buf.items():
            setattr(out, k + "_dev", t.data_ptr())
        if forced_actions is not None:
            mode = _lib.JN_MODE_FORCED
            forced_actions = forced_actions.to(dev, torch.int64).contiguous()
        else:
            mode = _lib.JN_MODE_SAMPLE if sample_actions else _lib.JN_MODE_GREEDY
        if start_positions is not None:
            start_positions = start_positions.to(dev, torch.int64).contiguous()
        self._rollouts += 1
        self.iter_num = getattr(self, "iter_num", 0) + 1
        ga = int(getattr(self.config, "gradient_accumulation", 1))
        opts = JnTrainOpts(struct_size=C.sizeof(JnTrainOpts), reward_norm=int(bool(self.config.reward_norm)),
                           ret_mean=float(self.last_return_mean), ret_std=float(self.last_return_std),
                           entropy_weight=float(self.entropy_weight), loss_scale=1.0 / ga)
        metrics_dev = torch.zeros(8, **f32)
        seed = (self.seed * 1000003 + self._rollouts) & 0xFFFFFFFFFFFFFFFF
        stream = _lib.current_stream(dev)
        check(eng.lib.jn_reinforce_step(eng.handle, mode, ptr(forced_actions), ptr(start_positions), seed,
                                        int(stop_early), C.byref(opts), C.byref(out), ptr(metrics_dev), stream),
              "jn_reinforce_step")
        m = metrics_dev.cpu()
        S = int(m[5])
        if self.config.reward_norm:
            lm = buf["logit_masks"][:, :S].bool()
            self.last_return_values.append(buf["returns"][:, :S][lm].clone())
        # detector training step on the patches that hold boxes (+ negatives), src/reinforce.py:330-339
        detection = bool(getattr(self.config, "detection_enabled", False)) and self.yolox_model() is not None
        yolo_losses = {}
        if detection:
            patches_y, boxes_y = env.get_detection_batch(int(getattr(self.config, "detection_sample_neg", 1)))
            if getattr(self, "detection_augment", None) is not None:               # src/reinforce.py:332-333
                patches_y = self.detection_augment(patches_y)
            yolo_losses = self.yolox_model().loss_and_backward(patches_y, boxes_y, loss_scale=1.0 / ga)
        if optimizer_step and self.iter_num % ga == 0:
            # the ONE exchange step of the iteration: flat gradient all-reduce (RCCL over xGMI)
            from .dist import allreduce_gradients
            scale = allreduce_gradients(grads, grads.numel() if detection else self._optim_numel, process_group,
                                        timing=getattr(self, "allreduce_timing", None))
            lr = float(getattr(self.config, "learning_rate", 1e-4))
            object.__setattr__(model, "_last_lr", (lr, float(getattr(self.config, "yolo_lr", lr))))     # save_checkpoint's default
            check(eng.lib.jn_optimizer_step(eng.handle, lr, 0.01, 1.0, scale, stream), "jn_optimizer_step")
            if detection:                          # optim_yolox.step(), src/reinforce.py:348-350
                ylr = float(getattr(self.config, "yolo_lr", lr))
                check(eng.lib.jn_optimizer_step_group(eng.handle, 1, ylr, 0.01, 1.0, scale, stream), "jn_optimizer_step_group")
            grads.zero_()
            model.refresh_flat_params()
            if self.config.reward_norm:
                self._compute_last_returns_mean_std()
        self._last_train_buffers = buf
        res = {"action_loss": m[0], "entropy_loss": m[1], "loss": m[2], "returns": m[3], "episode_length": m[4], "steps": S}
        for k, v in yolo_losses.items():
            res["yolo_" + k] = v
        return res

    def compute_metrics(self, rollout: Dict[str, torch.Tensor], env: NeedleGeneralEnv = None):
        """The reference's metric dictionary (src/reinforce.py:217-265): REINFORCE loss with window-normalised returns,
        entropy bonus, mean return and episode length over the valid steps; with ``env``, the found-ratios and STOP usage
        of image 0.  (``train_iteration`` gets the same numbers from ``reinforce_loss_kernel``; this form is the one that
        carries the autograd graph of ``training_step``.)"""
        valid = rollout["logit_masks"]
        n_valid = valid.sum()
        returns = rollout["returns"]
        if self.config.reward_norm:
            self.last_return_values.append(returns[valid].detach().clone())      # feeds the window statistics
            returns = (returns - self.last_return_mean) / (self.last_return_std + 1e-8)

        def masked_mean(x):
            return (x * valid).sum() / n_valid
        action_loss = -masked_mean(rollout["logprobs"] * returns)
        entropy_loss = -masked_mean(rollout["entropies"])
        metrics = {
            "action_loss": action_loss,
            "entropy_loss": entropy_loss,
            "loss": action_loss + self.entropy_weight * entropy_loss,
            "returns": (rollout["rewards"] * valid).sum(dim=1).mean(),
            "episode_length": valid.sum(dim=1).float().mean(),
        }
        if env:
            found = env.prop_patches_found[0]
            metrics["prop_patches_found"] = found
            metrics["prop_bbox_found"] = env.prop_bboxes_found[0]
            if self.stop_enabled:
                stopped = env.terminated[0]
                metrics["stop_used"] = stopped.to(torch.float32)
                metrics["stop_misused"] = (stopped & (found < 1)).to(torch.float32)
        return metrics

    @torch.no_grad()
    def eval_on_batch(self, env: NeedleGeneralEnv, do_detection: bool = None, merge_bboxes: bool = None) -> Dict[str, torch.Tensor]:
        """``eval_on_sample`` of the reference (src/reinforce.py:424-497) without the plotting: greedy rollout,
        rollout metrics incl. the env's found-ratios, and — with detection — mAP-50 of the boxes found along the
        trajectories (moved to full-image coordinates, optionally merged) plus the detector's mAP on every patch that
        holds a box (`yolo_map`).  The reference evaluates one image at a time; any batch size works here (the
        per-image metrics it reads from index 0 stay index 0)."""
        from .detection import (compute_detection_metrics, merge_boxes_batched, patch_bboxes2full_image)
        cfg = self.config
        if do_detection is None:
            do_detection = bool(getattr(cfg, "detection_enabled", False))
        if merge_bboxes is None:
            merge_bboxes = bool(getattr(cfg, "merge_bboxes", False))
        ro = self.rollout(env, sample_actions=False, do_detection=do_detection)
        metrics = self.compute_metrics(ro, env)
        if do_detection:
            targets = env.get_detection_targets()
            offsets = ro["positions"][:, :, [1, 0]] * self.patch_size        # (y, x) grid -> (x, y) pixels
            preds = patch_bboxes2full_image(ro["bboxes"], offsets, ro["masks"])
            if merge_bboxes:
                preds = merge_boxes_batched(preds, target=False)
                targets = merge_boxes_batched(targets, target=True)
            metrics.update(compute_detection_metrics(preds, targets))
            patches, patch_targets = env.get_detection_batch(sample_neg=0)
            pred_bboxes, _, yolo_losses = self.yolox_model()(patches)
            for k, v in compute_detection_metrics(pred_bboxes, list(patch_targets)).items():
                metrics["yolo_" + k] = v
            for k, v in yolo_losses.items():
                metrics["yolo_" + k] = v
        return metrics
