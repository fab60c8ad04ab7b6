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
        detection = self.model.yolox is not None and bool(getattr(cfg, "detection_enabled", True))
        aug = getattr(self, "detection_augment", None)
        if aug is not None:
            B_, T_ = cur.shape
            tr["patches"] = aug(tr["patches"].flatten(0, 1)).view(B_, T_, *tr["patches"].shape[2:])
            tr["patches_yolox"] = aug(tr["patches_yolox"])
        res = self.train_step(tr["patches"], cur, ref_actions, tr["positions"], masks, optimizer_step=False)
        ga = int(getattr(cfg, "gradient_accumulation", 1))
        if detection:
            _, _, yolo = self.model.yolox(tr["patches_yolox"], tr["bboxes_yolox"], loss_scale=1.0)
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

    def train_step(self, patches, current_actions, next_actions, positions, masks, optimizer_step: bool = True,
                   process_group=None) -> Dict[str, torch.Tensor]:
        """model(patches, current_actions, classes=0, positions) -> CE vs next_actions -> backward -> AdamW."""
        model, dev = self.model, self.device
        model.sync_weights()
        eng = model.engine()
        grads = self._grad_arena()
        B, T = current_actions.shape
        f = lambda t, dt: t.to(dev, dt).contiguous()
        patches, cur, nxt = f(patches, torch.float32), f(current_actions, torch.int64), f(next_actions, torch.int64)
        pos = None if positions is None else f(positions, torch.int64)
        msk = f(masks, torch.uint8)
        logits = torch.empty((B, T, eng.cfg.n_actions), device=dev, dtype=torch.float32)
        metrics = torch.zeros(4, device=dev, dtype=torch.float32)
        stream = _lib.current_stream(dev)
        check(eng.lib.jn_supervised_step(eng.handle, ptr(patches), ptr(cur), ptr(nxt), ptr(pos), ptr(msk), B, T,
                                         self.stop_weight, ptr(logits), ptr(metrics), stream), "jn_supervised_step")
        self.iter_num += 1
        ga = int(getattr(self.config, "gradient_accumulation", 1))
        if optimizer_step and self.iter_num % ga == 0:
            from .dist import allreduce_gradients
            scale = allreduce_gradients(grads, self._optim_numel, process_group)
            lr = float(getattr(self.config, "learning_rate", 1e-4))
            # the supervised loop does not clip gradients (src/supervised.py:897-902)
            check(eng.lib.jn_optimizer_step(eng.handle, lr, 0.01, 0.0, scale, stream), "jn_optimizer_step")
            grads.zero_()
            self.model.refresh_flat_params()
        m = metrics.cpu()
        return {"loss": m[0], "action_loss": m[0], "action_accuracy": m[1], "episode_length": m[2], "logits": logits}
