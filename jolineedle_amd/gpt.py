"""``GPT`` — the decision model behind the reference's operator API
(src/models/gpt.py:143-562), computed by libjnroll.so.

The module owns the parameters under the reference's state-dict names (so reference
checkpoints load and ``configure_optimizers`` splits by the ``yolox`` prefix exactly as
gpt.py:547-562) but holds no torch compute graph: ``forward`` uploads the current
weights (when they changed) and calls ``jn_gpt_forward``.  Eval-mode numerics
(BatchNorm running statistics, dropout must be 0).
"""
import math
from copy import deepcopy

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr
from .engine import Engine, make_jn_config, GPT_ZOO
from .yolox import NeedleYOLOX


class _Slot(nn.Module):
    """Parameter container: structure only, no forward."""

    def forward(self, *a, **k):
        raise RuntimeError("parameter container of the HIP engine: compute goes through GPT.forward / jn_* calls")


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, is_buffer: bool):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        nxt = mod._modules.get(p)
        if nxt is None:
            nxt = _Slot()
            mod.add_module(p, nxt)
        mod = nxt
    if is_buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor))


class _ForwardGraph(torch.autograd.Function):
    """Graph node behind the logits of a train-mode ``GPT.forward`` (the supervised loop, src/supervised.py:863-868, 897):
    backward = ``jn_supervised_backward(d loss / d logits)``, then the engine's packed gradients are added to
    ``param.grad`` (reference layout).  ``final_emb`` leaves without a graph (the supervised loop never differentiates it)."""

    @staticmethod
    def forward(ctx, anchor, model, gen, logits, keep):
        ctx.model, ctx.gen, ctx.keep = model, gen, keep     # keep: the input buffers the engine's backward reads
        return logits.view_as(logits)

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        if ctx.gen != model._rollout_gen:
            raise RuntimeError("backward through a forward whose activations were overwritten by a later train-mode pass")
        eng, dev = model.engine(), model.device
        dlogits = dlogits.to(dev, torch.float32).contiguous()
        check(eng.lib.jn_supervised_backward(eng.handle, ptr(dlogits), _lib.current_stream(dev)), "jn_supervised_backward")
        model.publish_engine_grads()
        return None, None, None, None, None


# Registration order of the reference's module tree (src/models/gpt.py:162-321 for the top level and the transformer;
# the published YOLOX modules for the detector / patch encoder).  ``named_parameters()`` must walk the
# tensors in THIS order: ``torch.optim.AdamW.state_dict()`` keys its moments by position in the parameter list, so a
# checkpoint's "optimizer-gpt" / "optimizer-yolox" entries only mean the same thing on both sides if the lists agree.
# The engine's own table is in execution order (conv2|conv1 pairs, bottlenecks before conv3).
_REF_ORDER = [
    ["action_head", "positional_encoding", "decoder_token_pos_enc", "embed_class", "project_concat", "yolox",
     "gpt_backbone", "embed_fpn", "transformer"],
    ["backbone", "head", "upsample", "lateral_conv0", "C3_p4", "reduce_conv1", "C3_p3", "bu_conv2", "C3_n3", "bu_conv1", "C3_n4"],
    ["stem", "dark2", "dark3", "dark4", "dark5"],
    ["conv1", "conv2", "conv3", "m"], ["dconv", "pconv"], ["conv", "bn"],
    ["cls_convs", "reg_convs", "cls_preds", "reg_preds", "obj_preds", "stems"],
    ["wte", "wpe", "drop", "h", "ln_f"], ["ln_1", "attn", "ln_2", "mlp"], ["c_attn", "c_fc", "c_proj"],
    ["weight", "bias", "running_mean", "running_var", "num_batches_tracked", "inv_freq", "cached_penc"],
]
_REF_RANK = {tok: i for group in _REF_ORDER for i, tok in enumerate(group)}


def reference_order_key(name: str):
    """Sort key that puts state-dict names in the reference's ``named_parameters()`` / ``state_dict()`` order."""
    return tuple((0, int(t)) if t.isdigit() else (1, _REF_RANK.get(t, 99), t) for t in name.split("."))


class GPT(nn.Module):
    """GPT Language Model driving the glimpse agent (drop-in for src/models/gpt.py:GPT)."""

    @staticmethod
    def get_default_config():
        from .config import CfgNode
        C = CfgNode()
        C.model_type = "gpt"
        C.n_layer = C.n_head = C.n_embd = None
        C.block_size = None
        C.embd_pdrop = C.resid_pdrop = C.attn_pdrop = 0.1
        return C

    def __init__(self, config, max_batch: int = 64, device="cuda:0"):
        super().__init__()
        assert config.block_size is not None
        type_given = config.model_type is not None
        params_given = all(getattr(config, k, None) is not None for k in ("n_layer", "n_head", "n_embd"))
        assert type_given ^ params_given          # gpt.py:181-189
        self.config = config
        self.block_size = config.block_size
        self.patch_size = config.patch_size
        self.image_processor = config.image_processor
        self.use_pos_emb, self.no_patch_emb = config.use_pos_emb, config.no_patch_emb
        self.concat_emb, self.decoder_pos_encoding = config.concat_emb, config.decoder_pos_encoding
        self.token_offset = 1
        self.device = torch.device(device)
        self.max_batch = max_batch
        n_actions = config.actions_info[0].nclasses
        dev_index = self.device.index or 0
        self._engine = Engine(make_jn_config(config, dev_index, max_batch, n_actions))
        self.n_layer, self.n_head, self.n_embd = (self._engine.cfg.n_layer, self._engine.cfg.n_head,
                                                 self._engine.cfg.n_embd)
        # --dropout (main.py:123-128; embd / attn / resid of the decision transformer): train-mode passes only
        self.dropout = float(getattr(config, "dropout", 0.0) or 0.0)
        self.set_dropout_seed(int(getattr(config, "seed", 0)))
        self._build_parameters()
        # --freeze-image-processor (gpt.py:264-268): yolox.backbone.* out of the optimiser
        self.freeze_image_processor = bool(getattr(config, "freeze_image_processor", False))
        if self.freeze_image_processor and self._engine.cfg.with_detector:
            check(self._engine.lib.jn_set_freeze(self._engine.handle, 1), "jn_set_freeze")
            for pn, p in self.named_parameters():
                if pn.startswith("yolox.backbone."):
                    p.requires_grad_(False)
        self._uploaded_version = None
        self._engine_grads = None          # packed gradient arena (torch-owned: ONE RCCL all-reduce per optimiser step)
        self._flat_params = self._flat_grads = None   # reference-layout mirrors behind param.data / param.grad (bind_flat)
        self._rollout_gen = 0
        if self._engine.cfg.with_detector:
            object.__setattr__(self, "_yolox_view", NeedleYOLOX(self, config.detector_conf_threshold))
        # Deviation from nn.Module's default: the model starts in eval mode (BatchNorm running statistics), the mode
        # every compute entry point but the training step uses; the reference's loops call model.train() / .eval()
        # themselves (src/reinforce.py:304, 366-370) and get the same modes here.
        self.eval()

    # ---- parameters under the reference's names -------------------------------------
    def _build_parameters(self):
        C = self.n_embd
        for name, shape, dtype, is_buffer, _used in sorted(self._engine.param_table(), key=lambda e: reference_order_key(e[0])):
            leaf = name.rsplit(".", 1)[-1]
            if dtype == 1:
                t = torch.zeros(shape, dtype=torch.long)
            elif name.endswith("inv_freq"):
                n = shape[0]
                ch = 2 * n
                t = 1.0 / (10000 ** (torch.arange(0, ch, 2).float() / ch))
            elif name.endswith(".attn.bias"):
                bs = shape[-1]
                t = torch.tril(torch.ones(bs, bs)).view(1, 1, bs, bs)
            elif ".bn." in name:
                t = torch.ones(shape) if leaf in ("weight", "running_var") else torch.zeros(shape)
            elif len(shape) == 4:                               # Conv2d default init
                t = torch.empty(shape)
                nn.init.kaiming_uniform_(t, a=math.sqrt(5))
            elif ("ln_" in name) and leaf == "weight":
                t = torch.ones(shape)
            elif leaf == "bias":
                t = torch.zeros(shape)
                if "cls_preds" in name or "obj_preds" in name:  # YOLOXHead.initialize_biases(1e-2)
                    t.fill_(-math.log((1 - 1e-2) / 1e-2))
            else:                                               # Linear / Embedding, gpt.py:536-545
                t = torch.empty(shape).normal_(0.0, 0.02)
                if name.endswith("c_proj.weight"):
                    t.normal_(0.0, 0.02 / math.sqrt(2 * self.n_layer))
            _attach(self, name, t, is_buffer)
        if self.decoder_pos_encoding and "wpe" in self.transformer._modules:
            self.transformer.wpe.weight.requires_grad_(False)   # gpt.py:320-321

    def set_dropout_seed(self, seed: int):
        """(Re)start the dropout mask stream: the n-th train-mode forward after this call uses `seed + n`."""
        check(self._engine.lib.jn_set_dropout(self._engine.handle, self.dropout, int(seed) & 0xFFFFFFFFFFFFFFFF), "jn_set_dropout")

    @property
    def yolox(self):
        return self._yolox_view

    def engine(self) -> Engine:
        return self._engine

    def _weights_version(self):
        return tuple(int(t._version) for t in self.state_dict(keep_vars=True).values())

    def sync_weights(self, force: bool = False):
        """Upload the current parameters (BN folded, Linear transposed) if they changed."""
        v = self._weights_version()
        if force or v != self._uploaded_version:
            self._engine.load_state_dict(self.state_dict())
            self._uploaded_version = v

    def load_state_dict(self, state_dict, strict=True):
        # main.py:541-543 strips DDP's "module." prefix before loading
        state_dict = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        out = super().load_state_dict(state_dict, strict=strict)
        self._uploaded_version = None
        return out

    # ---- forward ----------------------------------------------------------------------
    def forward(self, patches, actions, classes, positions=None, prev_embeddings=None):
        """Same contract as src/models/gpt.py:481-534.  ``classes`` [B] int64 selects the row of ``embed_class`` behind each
        sequence's class token (gpt.py:476-478; the RL loop passes zeros, src/reinforce.py:128-129, the supervised loop the
        batch's class ids, src/supervised.py:852, 866); ``None`` = class 0.  Eval mode / no-grad: BatchNorm running statistics, no graph.  ``model.train()`` + grad
        mode on a full sequence (no ``prev_embeddings``: the supervised loop's call, src/supervised.py:863-868): train-mode
        numerics — BatchNorm statistics over the B*T patches, running statistics updated, dropout — and the logits carry a
        graph whose backward is the engine's teacher-forced backward (``jn_supervised_forward`` / ``_backward``)."""
        seq_len = actions.shape[1]
        assert seq_len <= self.block_size, \
            f"Cannot forward sequence of length {seq_len}, block size is only {self.block_size}"
        assert (not self.use_pos_emb) or (positions is not None)
        assert actions.dim() == 2, "single categorical action only"
        if getattr(self.config, "no_recurrent_embedding", False):
            prev_embeddings = None
        self.sync_weights()
        dev = self.device
        B = actions.shape[0]
        C, nA = self.n_embd, self._engine.cfg.n_actions
        patches = None if self.no_patch_emb else patches.to(dev, torch.float32).contiguous()
        actions = actions.to(dev, torch.int64).contiguous()
        positions = None if positions is None else positions.to(dev, torch.int64).contiguous()
        if classes is not None:
            classes = torch.as_tensor(classes)
            assert classes.shape == (B,), f"classes must be [B] = [{B}], got {tuple(classes.shape)}"
            if classes.device.type == "cpu":          # nn.Embedding raises on an id outside its table; checked where it is free
                assert int(classes.min()) >= 0 and int(classes.max()) < 100, "class id outside embed_class (100 rows)"
            classes = classes.to(dev, torch.int64).contiguous()
        if self.training and torch.is_grad_enabled() and prev_embeddings is None and not self.no_patch_emb:
            assert B * seq_len <= self.max_batch, \
                f"train-mode forward: B*T = {B * seq_len} patches exceed max_batch = {self.max_batch} (BatchNorm statistics need one pass)"
            self.bind_flat()
            self._rollout_gen += 1
            logits = torch.empty((B, seq_len, nA), device=dev, dtype=torch.float32)
            final_emb = torch.empty((B, seq_len + 1, C), device=dev, dtype=torch.float32)
            check(self._engine.lib.jn_supervised_forward(self._engine.handle, ptr(patches), ptr(actions), ptr(classes),
                                                         ptr(positions), B, seq_len, ptr(logits), ptr(final_emb),
                                                         _lib.current_stream(dev)), "jn_supervised_forward")
            anchor = next(p for p in self.parameters() if p.requires_grad)
            logits = _ForwardGraph.apply(anchor, self, self._rollout_gen, logits, (patches, actions, classes, positions))
            return logits, final_emb
        Tp = 0
        if prev_embeddings is not None:
            prev_embeddings = prev_embeddings.to(dev, torch.float32).contiguous()
            Tp = prev_embeddings.shape[1]
        L = Tp + 1 if prev_embeddings is not None else seq_len + 1
        logits = torch.empty((B, L - 1, nA), device=dev, dtype=torch.float32)
        final_emb = torch.empty((B, L, C), device=dev, dtype=torch.float32)
        lib = self._engine.lib
        check(lib.jn_gpt_forward(self._engine.handle, ptr(patches), ptr(actions), ptr(classes), ptr(positions),
                                 ptr(prev_embeddings), B, seq_len, Tp, ptr(logits), ptr(final_emb),
                                 _lib.current_stream(dev)), "jn_gpt_forward")
        return logits, final_emb

    def embed_patches(self, patches):
        """[B, T, 3, P, P] -> [B, T, n_embd] (src/models/gpt.py:356-384)."""
        self.sync_weights()
        B, T = patches.shape[:2]
        flat = patches.to(self.device, torch.float32).reshape(B * T, *patches.shape[2:]).contiguous()
        out = torch.empty((B * T, self.n_embd), device=self.device, dtype=torch.float32)
        lib = self._engine.lib
        for i in range(0, B * T, self.max_batch):
            n = min(self.max_batch, B * T - i)
            check(lib.jn_embed_patches(self._engine.handle, ptr(flat[i:i + n]), n, ptr(out[i:i + n]),
                                       _lib.current_stream(self.device)), "jn_embed_patches")
        return out.view(B, T, -1)

    def pull_bn_statistics(self):
        """Copy the BatchNorm running statistics the engine updated in train mode back into this
        module's buffers (so state_dict() / checkpoints see them)."""
        eng = self._engine
        for name, buf in self.named_buffers():
            if name.endswith("running_mean") or name.endswith("running_var"):
                host = torch.empty_like(buf, device="cpu")
                check(eng.lib.jn_read_tensor(eng.handle, name.encode(), host.data_ptr(), host.numel()), "jn_read_tensor")
                buf.data.copy_(host)
        self._uploaded_version = self._weights_version()

    def backbone_features(self, patches, net=None, train=False):
        """fpn_outs of the patch encoder (gpt.py:375) or of the detector backbone, NCHW.
        train=True: batch-statistics BatchNorm over the given patches (one batch <= max_batch)."""
        self.sync_weights()
        cfg = self._engine.cfg
        if net is None:
            net = _lib.JN_NET_GPT_BACKBONE if cfg.gpt_bb_width > 0 else _lib.JN_NET_DETECTOR
        width = cfg.gpt_bb_width if net == _lib.JN_NET_GPT_BACKBONE else cfg.det_width
        N, P = patches.shape[0], self.patch_size
        x = patches.to(self.device, torch.float32).contiguous()
        chans = [int(256 * width), int(512 * width), int(1024 * width)]
        outs = [torch.empty((N, c, P // s, P // s), device=self.device) for c, s in zip(chans, (8, 16, 32))]
        lib = self._engine.lib
        assert not train or N <= self.max_batch, "train-mode statistics need the whole batch in one pass"
        for i in range(0, N, self.max_batch):
            n = min(self.max_batch, N - i)
            check(lib.jn_backbone_forward(self._engine.handle, net, ptr(x[i:i + n]), n, int(train), ptr(outs[0][i:i + n]),
                                          ptr(outs[1][i:i + n]), ptr(outs[2][i:i + n]),
                                          _lib.current_stream(self.device)), "jn_backbone_forward")
        return tuple(outs)

    # ---- training hooks (engine-side gradients) -------------------------------------------
    def engine_zero_grad(self):
        self.sync_weights()
        eng = self._engine
        check(eng.lib.jn_zero_grad(eng.handle, _lib.current_stream(self.device)), "jn_zero_grad")

    def backbone_backward(self, patches, grads, net=None):
        """Backward of the last ``backbone_features(patches, train=True)``: `grads` = dL/d(fpn_outs)
        (NCHW, entries may be None).  Parameter gradients accumulate inside the engine."""
        cfg = self._engine.cfg
        if net is None:
            net = _lib.JN_NET_GPT_BACKBONE if cfg.gpt_bb_width > 0 else _lib.JN_NET_DETECTOR
        x = patches.to(self.device, torch.float32).contiguous()
        gs = [None if g is None else g.to(self.device, torch.float32).contiguous() for g in grads]
        eng = self._engine
        check(eng.lib.jn_backbone_backward(eng.handle, net, ptr(x), x.shape[0], ptr(gs[0]), ptr(gs[1]), ptr(gs[2]),
                                           _lib.current_stream(self.device)), "jn_backbone_backward")

    def pull_parameters(self):
        """Copy the engine's (optimiser-updated) parameters and BN statistics back into this module."""
        eng = self._engine
        with torch.no_grad():
            for name, p in self.named_parameters():
                if not p.requires_grad:
                    continue
                host = torch.empty(p.shape, dtype=torch.float32)
                check(eng.lib.jn_read_param(eng.handle, name.encode(), host.data_ptr(), host.numel()), "jn_read_param")
                p.copy_(host)
        self.pull_bn_statistics()

    def engine_grads(self, prefix=""):
        """{state-dict name: gradient tensor (reference layout)} of the trainable tensors."""
        eng, out = self._engine, {}
        for name, p in self.named_parameters():
            if not name.startswith(prefix) or not p.requires_grad:
                continue
            host = torch.empty(p.shape, dtype=torch.float32)
            check(eng.lib.jn_read_grad(eng.handle, name.encode(), host.data_ptr(), host.numel()), "jn_read_grad")
            out[name] = host
        return out

    # ---- gradient arena and the reference-layout mirrors (autograd bridge) ------------------------------
    def grad_arena(self):
        """The engine's flat PACKED gradient arena as a torch tensor, owned by this model (one per engine, shared by every
        trainer): torch-owned so that RCCL all-reduces it in ONE call per optimiser step."""
        if self._engine_grads is None:
            import ctypes as C
            self.sync_weights()
            eng = self._engine
            tot, gpt = C.c_size_t(), C.c_size_t()
            check(eng.lib.jn_arena_info(eng.handle, C.byref(tot), C.byref(gpt)), "jn_arena_info")
            self._arena_numel, self._optim_gpt_numel = tot.value, gpt.value
            self._engine_grads = torch.zeros(tot.value, device=self.device, dtype=torch.float32)
            check(eng.lib.jn_set_grad_arena(eng.handle, ptr(self._engine_grads), tot.value), "jn_set_grad_arena")
        return self._engine_grads

    def bind_flat(self):
        """Make every trainable parameter a view of ONE device buffer in the reference's layout, with ``param.grad`` a view of
        a second one (same offsets as the engine's arena, jn_arena_segment).  After this, torch code sees real tensors:
        ``clip_grad_value_(model.parameters(), 1)`` clamps the gradients the engine optimiser will use, ``state_dict()`` is
        current after every ``optimizer.step()``.  Idempotent."""
        if self._flat_params is not None:
            return
        import ctypes as C
        self.grad_arena()
        eng, stream = self._engine, _lib.current_stream(self.device)
        n = self._arena_numel
        self._flat_params = torch.zeros(n, device=self.device, dtype=torch.float32)
        self._flat_grads = torch.zeros(n, device=self.device, dtype=torch.float32)
        check(eng.lib.jn_export_arena(eng.handle, 0, ptr(self._flat_params), n, 0, stream), "jn_export_arena")
        self._flat_names, self._flat_offsets = [], {}
        for name, p in self.named_parameters():
            off, num = C.c_size_t(), C.c_size_t()
            if eng.lib.jn_arena_segment(eng.handle, name.encode(), C.byref(off), C.byref(num)) != 0:
                continue                                  # not trainable in this configuration (e.g. wpe with sinusoid positions)
            assert num.value == p.numel(), name
            p.data = self._flat_params[off.value:off.value + num.value].view(p.shape)
            p.grad = self._flat_grads[off.value:off.value + num.value].view(p.shape)
            self._flat_names.append(name)
            self._flat_offsets[name] = off.value
        self._uploaded_version = self._weights_version()

    def refresh_flat_params(self):
        """After an optimiser step taken inside the engine (``train_iteration``): bring ``param.data`` of a bound model up to
        date.  No-op for an unbound model (its parameters are pulled on demand, ``pull_parameters``)."""
        if self._flat_params is None:
            return
        eng, stream = self._engine, _lib.current_stream(self.device)
        check(eng.lib.jn_export_arena(eng.handle, 0, ptr(self._flat_params), self._arena_numel, 0, stream), "jn_export_arena")
        self._uploaded_version = self._weights_version()

    def publish_engine_grads(self):
        """param.grad += the engine's (packed) gradient arena, then clear the arena: called after every engine backward
        when the model is bound (bind_flat), so that ``.grad`` accumulates like torch's."""
        eng, stream = self._engine, _lib.current_stream(self.device)
        check(eng.lib.jn_export_arena(eng.handle, 1, ptr(self._flat_grads), self._arena_numel, 1, stream), "jn_export_arena")
        self._engine_grads.zero_()

    def configure_optimizers(self, train_config):
        """gpt.py:547-562: everything not under ``yolox`` vs the detector — as ``torch.optim.Optimizer`` objects whose
        ``step()`` / ``zero_grad()`` drive the engine's AdamW on ``param.grad`` (EngineAdamW)."""
        from .optim import EngineAdamW
        gpt_params = [p for pn, p in self.named_parameters() if not pn.startswith("yolox")]
        optim_gpt = EngineAdamW(self, 0, gpt_params, lr=train_config.learning_rate)
        yolo_params = [p for pn, p in self.named_parameters() if pn.startswith("yolox")]
        optim_yolox = EngineAdamW(self, 1, yolo_params, lr=getattr(train_config, "yolo_lr", train_config.learning_rate)) \
            if yolo_params else None
        object.__setattr__(self, "_last_optimizers", (optim_gpt, optim_yolox))     # save_checkpoint's default
        return optim_gpt, optim_yolox
