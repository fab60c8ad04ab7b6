"""``SupervisedTrainer`` — the teacher-forced training step of the reference
(src/supervised.py:138-177 loss, :863-902 step) over ``jn_supervised_step``.  Trajectory generation
(NeedleSimpleEnv), augmentation, the detector loss and the evaluation suite are out of scope
(SURVEY.md §8): the caller supplies patches / actions / positions / masks of a batch."""
import ctypes as C
from typing import Dict

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
        if self._flat_grads is None:
            eng = self.model.engine()
            tot, gpt = C.c_size_t(), C.c_size_t()
            check(eng.lib.jn_arena_info(eng.handle, C.byref(tot), C.byref(gpt)), "jn_arena_info")
            self._flat_grads = torch.zeros(tot.value, device=self.device, dtype=torch.float32)
            self._optim_numel = gpt.value
            check(eng.lib.jn_set_grad_arena(eng.handle, ptr(self._flat_grads), tot.value), "jn_set_grad_arena")
        return self._flat_grads

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
        m = metrics.cpu()
        return {"loss": m[0], "action_loss": m[0], "action_accuracy": m[1], "episode_length": m[2], "logits": logits}
