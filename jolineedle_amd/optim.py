"""``torch.optim.Optimizer`` front of the engine's AdamW (src/models/gpt.py:547-562 returns two ``torch.optim.AdamW``; the
reference loop calls ``clip_grad_value_`` -> ``optim.step()`` -> ``optim.zero_grad()``, src/reinforce.py:344-350).

``step()`` takes ``param.grad`` (reference layout, views of the model's flat gradient buffer — already clipped by the caller
or not), averages it over the ranks with ONE all-reduce (data-parallel training; the reference itself has no gradient
exchange in RL mode), converts it to the arena layout on the device, runs ``jn_optimizer_step_group`` (AdamW, torch
defaults: betas 0.9 / 0.999, eps 1e-8, decoupled weight decay 0.01) and writes the new parameters back into
``param.data``.  The moments live in the engine.
"""
import torch

from . import _lib
from ._lib import check, ptr

_NO_CLIP = 0.0          # jn_optimizer_step_group: clip_value <= 0 = no clipping (the caller clips param.grad itself)


class EngineAdamW(torch.optim.Optimizer):
    def __init__(self, model, group: int, params, lr: float, weight_decay: float = 0.01, process_group=None,
                 sync_gradients: bool = True):
        params = list(params)
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay, betas=(0.9, 0.999), eps=1e-8))
        self._model, self._group, self._pg = model, int(group), process_group
        self.sync_gradients = sync_gradients      # False: the caller has already averaged param.grad over the ranks

    def _bound_range(self):
        """(lo, hi) of this group in the flat buffers; binds the model on first use (needs the GPU, so not in __init__:
        optimisers can be constructed on a host without one)."""
        m = self._model
        m.bind_flat()
        n_gpt, n_all = m._optim_gpt_numel, m._arena_numel
        return (0, n_gpt) if self._group == 0 else (n_gpt, n_all)

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None, "closures are not supported"
        m = self._model
        lo, hi = self._bound_range()
        eng, stream = m.engine(), _lib.current_stream(m.device)
        from .dist import allreduce_gradients
        scale = 1.0
        if hi > lo and self.sync_gradients:
            scale = allreduce_gradients(m._flat_grads[lo:hi], hi - lo, self._pg)
        # reference layout -> arena; only this group's range is consumed by the kernel below
        check(eng.lib.jn_import_arena(eng.handle, 1, ptr(m._flat_grads), m._arena_numel, stream), "jn_import_arena")
        g = self.param_groups[0]
        check(eng.lib.jn_optimizer_step_group(eng.handle, self._group, float(g["lr"]), float(g["weight_decay"]), _NO_CLIP,
                                              float(scale), stream), "jn_optimizer_step_group")
        m._engine_grads.zero_()
        check(eng.lib.jn_export_arena(eng.handle, 0, ptr(m._flat_params), m._arena_numel, 0, stream), "jn_export_arena")
        m._uploaded_version = m._weights_version()        # the engine already holds these values
        return None

    def zero_grad(self, set_to_none: bool = False):
        """Zeroes this group's slice of the flat gradient buffer in place (``param.grad`` stays a view of it)."""
        lo, hi = self._bound_range()
        if hi > lo:
            self._model._flat_grads[lo:hi].zero_()
