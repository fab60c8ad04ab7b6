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
        # torch-side edits of a bound model (load_state_dict, manual in-place changes) reach the engine before the update:
        # the export below would otherwise overwrite them and mark them as uploaded
        m.sync_weights()
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

    # ---- checkpoint state in torch.optim.AdamW's layout (main.py:436-449, 532-563: "optimizer-gpt" / "optimizer-yolox") ----
    def _moments(self, what: int) -> torch.Tensor:
        m = self._model
        buf = torch.zeros(m._arena_numel, device=m.device, dtype=torch.float32)
        check(m.engine().lib.jn_export_arena(m.engine().handle, what, ptr(buf), buf.numel(), 0, _lib.current_stream(m.device)), "jn_export_arena")
        return buf

    def state_dict(self):
        """{"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]} over this optimiser's parameters in order,
        the moments converted from the engine's arena to the reference layout: a dict ``torch.optim.AdamW.load_state_dict``
        accepts for the same parameter list."""
        import ctypes as C
        m = self._model
        m.bind_flat()
        steps = C.c_int()
        check(m.engine().lib.jn_optimizer_steps(m.engine().handle, self._group, C.byref(steps), 0), "jn_optimizer_steps")
        ea, eas = self._moments(2), self._moments(3)
        names = {id(p): n for n, p in m.named_parameters()}
        state, idx = {}, []
        for i, p in enumerate(self.param_groups[0]["params"]):
            idx.append(i)
            n = names.get(id(p))
            off = m._flat_offsets.get(n) if n is not None else None
            if off is None or not p.requires_grad or steps.value == 0:
                continue
            sl = slice(off, off + p.numel())
            state[i] = {"step": torch.tensor(float(steps.value)), "exp_avg": ea[sl].view(p.shape).cpu().clone(),
                        "exp_avg_sq": eas[sl].view(p.shape).cpu().clone()}
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        g.update(params=idx, amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
        # which tensor stands behind every position: torch keys optimizer state by position in the parameter list, and this
        # package's list order changed once (engine execution order -> the reference's module order, round 3); with the
        # names in the file a reader can map by NAME whatever its own order is (torch >= 2.6 writes the same key for
        # optimisers built from named_parameters(); torch.optim.AdamW.load_state_dict ignores it otherwise)
        g["param_names"] = [names.get(id(p), "") for p in self.param_groups[0]["params"]]
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd):
        """Accepts what ``state_dict`` writes and what ``torch.optim.AdamW.state_dict()`` of the reference writes for the same
        parameter list (main.py:548-556); an empty dict (a checkpoint of round 1) leaves the moments at zero."""
        import ctypes as C
        if not sd or not sd.get("state"):
            return
        m = self._model
        m.bind_flat()
        ea, eas = self._moments(2), self._moments(3)
        names = {id(p): n for n, p in m.named_parameters()}
        # position -> entry of the file.  With "param_names" in the file the entries are matched by NAME (any list order:
        # files of this package, of torch >= 2.6 optimisers built from named_parameters()).  Without it positions are all
        # there is: they are read in the reference's module order — what torch.optim.AdamW of the reference wrote for
        # GPT.configure_optimizers (src/models/gpt.py:547-562).  A file this package wrote BEFORE it carried names (rounds
        # 1-2: engine execution order, conv2|conv1 pairs first) cannot be told from that by content — same shapes inside a
        # CSP pair — so the untagged case is announced instead of silently trusted.
        file_names = (sd.get("param_groups") or [{}])[0].get("param_names")
        by_name = None
        if file_names:
            ids = (sd.get("param_groups") or [{}])[0].get("params", list(range(len(file_names))))
            by_name = {n: sd["state"].get(i, sd["state"].get(str(i))) for n, i in zip(file_names, ids)}
            unknown = sorted(n for n in by_name if n and n not in names.values() and by_name[n] is not None)
            if unknown:
                raise ValueError(f"optimizer state names tensors this model does not have: {unknown[:4]} ...")
        else:
            import warnings
            warnings.warn("optimizer state without 'param_names': positions are read in the reference's parameter order "
                          "(torch.optim.AdamW over GPT.named_parameters()); a file written by rounds 1-2 of this package "
                          "(engine execution order) must be re-saved with names, its moments would land on the wrong tensors")
        step = 0
        for i, p in enumerate(self.param_groups[0]["params"]):
            st = by_name.get(names.get(id(p))) if by_name is not None else sd["state"].get(i, sd["state"].get(str(i)))
            off = m._flat_offsets.get(names.get(id(p)))
            if st is None or off is None:
                continue
            # positions mean the same tensor on both sides only if the parameter lists agree (they do for the
            # reference's module order, gpt.py::reference_order_key); a foreign list is skipped entry by entry, not crashed on
            if tuple(st["exp_avg"].shape) != tuple(p.shape) or tuple(st["exp_avg_sq"].shape) != tuple(p.shape):
                import warnings
                warnings.warn(f"optimizer state {i} has shape {tuple(st['exp_avg'].shape)}, parameter {names.get(id(p))} "
                              f"has {tuple(p.shape)}: entry skipped (moments stay as they are)")
                continue
            sl = slice(off, off + p.numel())
            ea[sl] = st["exp_avg"].to(m.device, torch.float32).flatten()
            eas[sl] = st["exp_avg_sq"].to(m.device, torch.float32).flatten()
            step = max(step, int(float(st["step"])))
        eng, stream = m.engine(), _lib.current_stream(m.device)
        # (both groups share the arena: the other group's slices were exported above and go back unchanged)
        check(eng.lib.jn_import_arena(eng.handle, 2, ptr(ea), ea.numel(), stream), "jn_import_arena")
        check(eng.lib.jn_import_arena(eng.handle, 3, ptr(eas), eas.numel(), stream), "jn_import_arena")
        steps = C.c_int(step)
        check(eng.lib.jn_optimizer_steps(eng.handle, self._group, C.byref(steps), 1), "jn_optimizer_steps")
        if sd.get("param_groups"):
            # torch.optim.AdamW.load_state_dict restores the hyper-parameters too (and the reference's later `optim.lr = lr`
            # is a no-op, main.py:548-556).  A non-positive learning rate is never a training value: such an entry (a
            # checkpoint written without optimisers by an earlier version of save_checkpoint) keeps the configured one.
            for k in ("lr", "weight_decay"):
                if k in sd["param_groups"][0] and (k != "lr" or float(sd["param_groups"][0][k]) > 0.0):
                    self.param_groups[0][k] = sd["param_groups"][0][k]

    def zero_grad(self, set_to_none: bool = False):
        """Zeroes this group's slice of the flat gradient buffer in place (``param.grad`` stays a view of it)."""
        lo, hi = self._bound_range()
        if hi > lo:
            self._model._flat_grads[lo:hi].zero_()
