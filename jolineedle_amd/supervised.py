"""``SupervisedTrainer`` — the teacher-forced training step of the reference
(src/supervised.py:138-177 loss, :863-902 step) over ``jn_supervised_step``, fed by teacher trajectories
(``generate_trajectories``, src/supervised.py:95-136, over trajectory.NeedleSimpleEnv) whose patches are gathered on
the device, and followed by the detector step on the trajectories' detector patches (src/supervised.py:881-902).
The evaluation suite is out of scope; augmentation is opt-in (``init_detection``)."""
import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import check, ptr


class SupervisedTrainer:
    def __init__(self, config, model, logger=None, train_dataset=None, test_dataset=None, rank: int = 0):
        self.config, self.model, self.rank = config, model, rank
        self.logger, self.train_dataset, self.test_dataset = logger, train_dataset, test_dataset
        self.device = model.device
        self.stop_weight = float(getattr(config, "stop_weight", 1.0)) if getattr(config, "stop_enabled", False) else 1.0
        self.best_metric_name = "map"
        self.iter_num = 0
        self._flat_grads = None

    def _grad_arena(self):
        """The engine's flat fp32 gradient arena: owned by the MODEL (one per engine, shared by all trainers)."""
        g = self.model.grad_arena()
        self._optim_numel = self.model._optim_gpt_numel
        return g

    def init_detection(self, **kw):
        """``Trainer.init_detection`` (src/trainer.py:176-186): on-device augmentation of the trajectory patches and the
        detector patches (src/supervised.py:855-861, 884-885).  Off until called."""
        from .augment import DetectionAugment
        kw.setdefault("seed", int(getattr(self.config, "seed", 0)) + 17 * int(self.rank))
        self.detection_augment = DetectionAugment(**kw)
        return self.detection_augment

    def generate_trajectories(self, batch: Dict, position: Optional[Tuple[int, int]] = None, seed: Optional[int] = None) -> Dict:
        """One teacher walk per image of `batch` (``image`` [B, C, H, W] on the device or a list of [C, H, W] of one
        size, ``bboxes`` [B, nb, 4] xyxy with zero-row padding or a list, ``class_id``) -> the collated dict of
        src/supervised.py:95-136: patches [B, T, C, P, P], current_actions / next_actions / labels [B, T],
        positions [B, T, 2], masks [B, T], local_bboxes [B, T, nb, 6], patches_yolox [M, C, P, P], bboxes_yolox
        [M, nb, 6], class_id [B].  The walks are integer work on the host; both patch tensors come from the
        device-resident images in two gather launches.  `seed` makes the walks reproducible (the reference seeds
        nothing here)."""
        from .trajectory import NeedleSimpleEnv, assemble_samples
        cfg = self.config
        images = batch["image"]
        if not isinstance(images, torch.Tensor):
            images = torch.stack(list(images))
        images = images.to(self.device, torch.float32).contiguous()
        P = int(cfg.patch_size)
        idx = []
        for i in range(images.shape[0]):
            bb = batch["bboxes"][i]
            if isinstance(bb, torch.Tensor):
                bb = bb[(bb != 0).any(dim=-1)] if bb.numel() else bb.reshape(0, 4)      # drop the collate's zero rows
            env = NeedleSimpleEnv(None, P, bb, seed=None if seed is None else seed + i, height=images.shape[2], width=images.shape[3])
            idx.append(env.generate_sample_indices(int(cfg.max_seq_len), int(getattr(cfg, "min_keypoints", 0)),
                                                   int(getattr(cfg, "max_keypoints", 0)),
                                                   bool(getattr(cfg, "binomial_keypoints", False)), position))
        out = assemble_samples(images, list(range(images.shape[0])), idx, P)
        cid = batch.get("class_id")
        out["class_id"] = (torch.as_tensor(cid) if cid is not None else torch.zeros(images.shape[0], dtype=torch.long)).to(self.device, torch.long)
        return out

    def train_iteration(self, batch: Dict, optimizer_step: bool = True, process_group=None, seed: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """Loop body of src/supervised.py:844-902 without augmentation: trajectories -> teacher-forced step (CE loss,
        backward) -> detector loss + backward on the trajectories' detector patches -> AdamW on both groups."""
        cfg = self.config
        tr = self.generate_trajectories(batch, seed=seed)
        cur, nxt, masks = tr["current_actions"], tr["next_actions"], tr["masks"]
        if getattr(cfg, "loss_mode", "best-action") == "on-self-trajectory":          # src/supervised.py:870-877
            ref_actions = torch.zeros_like(cur)
            ref_actions[:, :-1] = cur[:, 1:]
            last = masks.sum(dim=1).long() - 1
            rows = torch.arange(cur.shape[0], device=cur.device)
            ref_actions[rows, last] = nxt[rows, last]
        else:
            ref_actions = nxt
        detection = self.yolox_model() is not None and bool(getattr(cfg, "detection_enabled", True))
        aug = getattr(self, "detection_augment", None)
        if aug is not None:
            B_, T_ = cur.shape
            tr["patches"] = aug(tr["patches"].flatten(0, 1)).view(B_, T_, *tr["patches"].shape[2:])
            tr["patches_yolox"] = aug(tr["patches_yolox"])
        res = self.train_step(tr["patches"], cur, ref_actions, tr["positions"], masks, optimizer_step=False, classes=tr["class_id"])
        ga = int(getattr(cfg, "gradient_accumulation", 1))
        if detection:
            yolo = self.model.yolox.loss_and_backward(tr["patches_yolox"], tr["bboxes_yolox"], loss_scale=1.0)
            for k, v in yolo.items():
                res["yolo_" + k] = v
            res["loss"] = res["loss"] + yolo["total_loss"].cpu()
        if optimizer_step and self.iter_num % ga == 0:
            from .dist import allreduce_gradients
            eng, grads = self.model.engine(), self._grad_arena()
            stream = _lib.current_stream(self.device)
            scale = allreduce_gradients(grads, grads.numel() if detection else self._optim_numel, process_group)
            lr = float(getattr(cfg, "learning_rate", 1e-4))
            check(eng.lib.jn_optimizer_step(eng.handle, lr, 0.01, 0.0, scale, stream), "jn_optimizer_step")
            if detection:
                ylr = float(getattr(cfg, "yolo_lr", lr))
                check(eng.lib.jn_optimizer_step_group(eng.handle, 1, ylr, 0.01, 0.0, scale, stream), "jn_optimizer_step_group")
            grads.zero_()
            self.model.refresh_flat_params()
        res["trajectories"] = tr
        return res

    # ---- the reference's supervised loop on the autograd bridge (src/supervised.py:138-198, 812-911) -------------------
    def yolox_model(self):
        return getattr(self.model, "_yolox_view", None)      # None: no detector configured (with_detector = False)

    def ddp_setup(self, rank: int, world_size: int, port: int, backend: str = None):
        """``Trainer.ddp_setup`` (src/trainer.py:61-71); rendezvous on 127.0.0.1, ``backend="gloo"`` for rehearsals."""
        import os
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        if not dist.is_initialized():
            dist.init_process_group(backend=backend or "nccl", rank=rank, world_size=world_size)
            self._owns_process_group = True           # run() only tears down a group it created itself

    def compute_metrics(self, action_logits, actions, masks, yolo_loss: Optional[dict] = None) -> Dict[str, torch.Tensor]:
        """src/supervised.py:138-198: CrossEntropy(weight[STOP] = stop_weight, reduction none) over the non-padding tokens,
        accuracy, the detector's loss terms under ``yolo_*`` and their total added to ``loss``, ``episode_length``."""
        nA = action_logits.shape[-1]
        weight = torch.ones(nA, device=action_logits.device)
        if getattr(self.config, "stop_enabled", False):
            weight[-1] = self.stop_weight
        flat, target = action_logits.reshape(-1, nA), actions.flatten()
        valid = (masks == 1).flatten()
        per_token = torch.nn.functional.cross_entropy(flat, target, weight=weight, reduction="none")
        metrics = {"action_loss": per_token[valid].mean(),
                   "action_accuracy": (target[valid] == flat.argmax(dim=1)[valid]).float().mean()}
        metrics["loss"] = metrics["action_loss"]
        if yolo_loss is not None:
            for k, v in yolo_loss.items():
                metrics["yolo_" + k] = v if isinstance(v, torch.Tensor) else torch.tensor(v)
            metrics["yolo_loss"] = metrics["yolo_total_loss"]
            metrics["loss"] = metrics["loss"] + metrics["yolo_loss"].to(metrics["loss"].device)
        if torch.isnan(metrics["action_accuracy"]):
            metrics["action_accuracy"] = torch.zeros((), device=action_logits.device)
        metrics["episode_length"] = masks.sum(dim=1).float().mean()
        return metrics

    def training_step(self, batch: Dict, optim_gpt, optim_yolox=None, seed: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """The body of the reference's loop (src/supervised.py:834-902): trajectories -> [augmentation] ->
        ``model(patches, current_actions, classes, positions)`` -> reference actions -> detector loss -> compute_metrics
        -> ``loss.backward()`` -> every ga-th iteration ``optim.step()`` / ``zero_grad()`` (no clipping).  ``model`` is this
        package's GPT in train mode: the logits carry a graph whose backward is the engine's, and so does the detector's
        ``total_loss`` (``yolox.py::_DetectorGraph``): ONE ``loss.backward()`` runs both; the optimisers average the gradients over the ranks with ONE all-reduce of the flat buffer each
        (the job DDP's bucketed all-reduce does in the reference, src/supervised.py:815)."""
        cfg, model = self.config, self.model
        self.iter_num += 1
        model.train()
        batch = self.generate_trajectories(batch, seed=seed)
        patches, current_actions, next_actions = batch["patches"], batch["current_actions"], batch["next_actions"]
        positions, masks, classes = batch["positions"], batch["masks"], batch["class_id"]
        aug = getattr(self, "detection_augment", None)
        if aug is not None:
            with torch.no_grad():
                B_, T_ = current_actions.shape
                patches = aug(patches.flatten(0, 1)).view(B_, T_, *patches.shape[2:])
        action_logits, _ = model(patches, current_actions, classes=classes, positions=positions)
        if getattr(cfg, "loss_mode", "best-action") == "on-self-trajectory":
            reference_actions = torch.zeros_like(current_actions)
            reference_actions[:, :-1] = current_actions[:, 1:]
            last = masks.sum(dim=1).long() - 1
            rows = torch.arange(current_actions.shape[0], device=current_actions.device)
            reference_actions[rows, last] = next_actions[rows, last]
        else:
            reference_actions = next_actions
        yolo_loss = None
        if self.yolox_model() is not None and bool(getattr(cfg, "detection_enabled", True)):
            patches_yolox = batch["patches_yolox"]
            if aug is not None:
                with torch.no_grad():
                    patches_yolox = aug(patches_yolox)
            _, _, yolo_loss = self.yolox_model()(patches_yolox, batch["bboxes_yolox"], predict=False)   # :881-888; total_loss has a graph
        metrics = self.compute_metrics(action_logits, reference_actions, masks, yolo_loss)
        metrics["loss"].backward()
        if self.iter_num % int(getattr(cfg, "gradient_accumulation", 1)) == 0:
            optim_gpt.step()
            if optim_yolox is not None:
                optim_yolox.step()
            optim_gpt.zero_grad()
            if optim_yolox is not None:
                optim_yolox.zero_grad()
        return {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in metrics.items()}

    def run(self, rank: int, world_size: int, port: int, batches=None, max_iters: int = None, backend: str = None,
            seed: Optional[int] = None):
        """``SupervisedTrainer.run`` (src/supervised.py:812-911) without the dataset / Visdom / test plumbing (out of scope,
        SURVEY.md §8): `batches` is any iterable of collated batches ({"image", "bboxes"[, "class_id"]}), re-iterated when
        exhausted.  One process per GPU; gradients are averaged by the optimisers' all-reduce (DDP semantics), BatchNorm
        statistics stay per rank.  Returns the metrics of the last iteration."""
        import torch.distributed as dist
        if bool(getattr(self.config, "augment_detection", False)):
            self.init_detection()
        self.rank = rank
        if world_size > 1 or backend:
            self.ddp_setup(rank, world_size, port, backend)
        optim_gpt, optim_yolox = self.model.configure_optimizers(self.config)
        for o in (optim_gpt, optim_yolox):
            if o is not None:
                o.sync_gradients = world_size > 1
        self.optim_gpt, self.optim_yolox = optim_gpt, optim_yolox
        batches = batches if batches is not None else getattr(self, "train_dataset", None)
        assert batches is not None, "run() needs an iterable of collated batches"
        n_iters = int(max_iters if max_iters is not None else getattr(self.config, "max_iters", 1))
        it, metrics = iter(batches), None
        for i in range(n_iters):
            try:
                batch = next(it)
            except StopIteration:
                it = iter(batches)
                batch = next(it)
            metrics = self.training_step(batch, optim_gpt, optim_yolox, seed=None if seed is None else seed + i)
        if getattr(self, "_owns_process_group", False) and dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()                 # src/reinforce.py:362 / src/supervised.py:911
            self._owns_process_group = False
        return metrics

    def train_step(self, patches, current_actions, next_actions, positions, masks, optimizer_step: bool = True,
                   process_group=None, classes=None) -> Dict[str, torch.Tensor]:
        """model(patches, current_actions, classes, positions) -> CE vs next_actions -> backward -> AdamW
        (``classes`` [B] class ids as in src/supervised.py:852, 866; None = class 0)."""
        model, dev = self.model, self.device
        model.sync_weights()
        eng = model.engine()
        grads = self._grad_arena()
        B, T = current_actions.shape
        f = lambda t, dt: t.to(dev, dt).contiguous()
        patches, cur, nxt = f(patches, torch.float32), f(current_actions, torch.int64), f(next_actions, torch.int64)
        pos = None if positions is None else f(positions, torch.int64)
        msk = f(masks, torch.uint8)
        cls = None if classes is None else f(torch.as_tensor(classes), torch.int64)
        assert cls is None or cls.shape == (B,), "classes must be [B]"
        logits = torch.empty((B, T, eng.cfg.n_actions), device=dev, dtype=torch.float32)
        metrics = torch.zeros(4, device=dev, dtype=torch.float32)
        stream = _lib.current_stream(dev)
        check(eng.lib.jn_supervised_step(eng.handle, ptr(patches), ptr(cur), ptr(nxt), ptr(cls), ptr(pos), ptr(msk), B, T,
                                         self.stop_weight, ptr(logits), ptr(metrics), stream), "jn_supervised_step")
        self.iter_num += 1
        ga = int(getattr(self.config, "gradient_accumulation", 1))
        if optimizer_step and self.iter_num % ga == 0:
            from .dist import allreduce_gradients
            scale = allreduce_gradients(grads, self._optim_numel, process_group)
            lr = float(getattr(self.config, "learning_rate", 1e-4))
            object.__setattr__(model, "_last_lr", (lr, float(getattr(self.config, "yolo_lr", lr))))     # save_checkpoint's default
            # the supervised loop does not clip gradients (src/supervised.py:897-902)
            check(eng.lib.jn_optimizer_step(eng.handle, lr, 0.01, 0.0, scale, stream), "jn_optimizer_step")
            grads.zero_()
            self.model.refresh_flat_params()
        m = metrics.cpu()
        return {"loss": m[0], "action_loss": m[0], "action_accuracy": m[1], "episode_length": m[2], "logits": logits}
