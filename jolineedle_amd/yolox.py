"""``NeedleYOLOX`` — detector wrapper of the reference (src/models/yolox.py:15-120): inference branch
(``jn_detect``) and the training loss branch (``jn_detector_forward`` / ``jn_detector_backward`` behind an autograd
node, or ``jn_detector_step`` in one call), computed by libjnroll.so."""
from typing import List, Optional

import torch

from . import _lib
from ._lib import check, ptr

LOSS_NAMES = ("total_loss", "iou_loss", "conf_loss", "cls_loss", "l1_loss", "num_fg")


class _DetectorGraph(torch.autograd.Function):
    """Graph node behind ``losses["total_loss"]`` of ``NeedleYOLOX.forward(patches, targets)`` in grad mode (SURVEY.md §8b;
    src/models/yolox.py:58-73 returns a differentiable loss, src/reinforce.py:336-341 adds it to the policy loss before ONE
    ``backward()``).  backward = ``jn_detector_backward`` per resident pass with the upstream scalar torch hands over, then the
    engine's packed gradients are added to ``param.grad`` (reference layout)."""

    @staticmethod
    def forward(ctx, anchor, model, gen, total_loss, passes, keep):
        # `keep`: the patch / target chunks the engine's backward reads through raw pointers (stem weight gradient)
        ctx.model, ctx.gen, ctx.passes, ctx.keep = model, gen, passes, keep
        return total_loss.view_as(total_loss)

    @staticmethod
    def backward(ctx, dloss):
        model = ctx.model
        if ctx.gen != model._detector_gen:
            raise RuntimeError("backward through a detector forward whose activations were overwritten by a later training pass")
        eng, dev = model.engine(), model.device
        dloss = dloss.to(dev, torch.float32).contiguous()
        for index, weight in ctx.passes:
            check(eng.lib.jn_detector_backward(eng.handle, index, ptr(dloss), float(weight), _lib.current_stream(dev)),
                  "jn_detector_backward")
        model.publish_engine_grads()
        return None, None, None, None, None, None


class NeedleYOLOX:
    """View on the ``yolox.*`` part of a GPT's engine.

    ``forward(patches, targets=None) -> (outputs, fpn_outs, losses)`` as src/models/yolox.py:24-91.

    Without ``targets``: the inference branch (eval-mode BatchNorm), ``losses == {}``.

    With ``targets`` ([N, nb, 5 | 6] = class, x1, y1, x2, y2[, 1]; zero rows = padding): the engine runs PAFPN + head in
    train mode (batch-statistics BatchNorm, running statistics updated), the SimOTA assignment and the IoU / objectness /
    class / L1 losses, and — as the reference does after its loss branch (:74-91) — the eval-mode head on the same FPN
    maps, postprocess and clamp: ``outputs`` are those predictions, ``fpn_outs`` the three (train-mode) maps.  In grad
    mode ``losses["total_loss"]`` carries a graph whose backward is the engine's: the reference's statements
    ``loss += yolo_loss["total_loss"]; (loss / ga).backward()`` (src/reinforce.py:339-341, src/supervised.py:888-897) run
    as written, in either order with the policy loss's own backward.  The other entries of ``losses`` are values
    (the reference's loops only log them).  ``predict=False`` skips the eval head / maps (the loops discard them).

    A detection batch larger than ``max_batch`` is fed in chunks of ``max_batch``, each weighted by its share of the
    patches (deviation: BatchNorm statistics and the 1 / num_fg normalisation are per chunk); every chunk keeps its own
    workspace until the backward.
    """

    def __init__(self, gpt, conf_threshold: float):
        self._gpt = gpt
        self.conf_threshold = conf_threshold

    def __call__(self, patches, targets=None, predict: bool = True):
        return self.forward(patches, targets, predict)

    def _boxes_to_list(self, boxes, counts) -> List[Optional[torch.Tensor]]:
        cnt = counts.tolist()
        return [boxes[i, :c].clone() if c > 0 else None for i, c in enumerate(cnt)]

    def forward(self, patches: torch.Tensor, targets: Optional[torch.Tensor] = None, predict: bool = True):
        g = self._gpt
        g.sync_weights()
        eng = g.engine()
        N = patches.shape[0]
        K = eng.cfg.max_det_per_patch
        x = patches.to(g.device, torch.float32).contiguous()
        if targets is None:
            boxes = torch.zeros((N, K, 7), device=g.device)
            counts = torch.zeros((N,), device=g.device, dtype=torch.int32)
            for i in range(0, N, g.max_batch):
                n = min(g.max_batch, N - i)
                check(eng.lib.jn_detect(eng.handle, ptr(x[i:i + n]), n, ptr(boxes[i:i + n]), ptr(counts[i:i + n]),
                                        None, _lib.current_stream(g.device)), "jn_detect")
            fpn_outs = g.backbone_features(x, _lib.JN_NET_DETECTOR)
            return self._boxes_to_list(boxes, counts), fpn_outs, {}
        # ---- loss branch + eval head (src/models/yolox.py:58-91) ----
        assert targets.shape[0] == N and targets.shape[-1] >= 5
        t = targets[..., :5].to(g.device, torch.float32).contiguous()
        graph = torch.is_grad_enabled()
        if graph:
            g.bind_flat()
        g._detector_gen = getattr(g, "_detector_gen", 0) + 1
        cfg = eng.cfg
        P = g.patch_size
        boxes = counts = None
        fpn = [None, None, None]
        if predict:
            boxes = torch.zeros((N, K, 7), device=g.device)
            counts = torch.zeros((N,), device=g.device, dtype=torch.int32)
            chans = [int(256 * cfg.det_width), int(512 * cfg.det_width), int(1024 * cfg.det_width)]
            fpn = [torch.empty((N, c, P // s, P // s), device=g.device) for c, s in zip(chans, (8, 16, 32))]
        tot = {k: torch.zeros((), device=g.device) for k in LOSS_NAMES}
        starts = list(range(0, N, g.max_batch))
        passes, keep = [], []
        for index, i in enumerate(starts):
            n = min(g.max_batch, N - i)
            metrics = torch.zeros(8, device=g.device)
            xc, tc = x[i:i + n], t[i:i + n]
            check(eng.lib.jn_detector_forward(eng.handle, ptr(xc), n, ptr(tc), t.shape[1], index, len(starts), ptr(metrics),
                                              ptr(boxes[i:i + n]) if predict else None, ptr(counts[i:i + n]) if predict else None,
                                              *(ptr(f[i:i + n]) if predict else None for f in fpn),
                                              _lib.current_stream(g.device)), "jn_detector_forward")
            for j, k in enumerate(LOSS_NAMES):
                tot[k] = tot[k] + metrics[j] * (n / N)
            passes.append((index, n / N))
            keep.append((xc, tc))
        if graph:
            anchor = next(p for n_, p in g.named_parameters() if n_.startswith("yolox") and p.requires_grad)
            tot["total_loss"] = _DetectorGraph.apply(anchor, g, g._detector_gen, tot["total_loss"], passes, (x, t, keep))
        outputs = self._boxes_to_list(boxes, counts) if predict else [None] * N
        return outputs, (tuple(fpn) if predict else None), tot

    def loss_and_backward(self, patches: torch.Tensor, targets: torch.Tensor, loss_scale: float = 1.0) -> dict:
        """Loss branch of src/models/yolox.py:58-73 AND ``(loss_scale * total_loss).backward()`` in one engine call
        (``jn_detector_step``): the fast path of ``train_iteration``.  Returns the reference's loss dict (device scalars, no
        graph); the yolox.* gradients are accumulated in the engine's gradient arena."""
        g = self._gpt
        g.sync_weights()
        eng = g.engine()
        N = patches.shape[0]
        assert targets.shape[0] == N and targets.shape[-1] >= 5
        x = patches.to(g.device, torch.float32).contiguous()
        t = targets[..., :5].to(g.device, torch.float32).contiguous()
        tot = {k: torch.zeros((), device=g.device) for k in LOSS_NAMES}
        for i in range(0, N, g.max_batch):
            n = min(g.max_batch, N - i)
            metrics = torch.zeros(8, device=g.device)
            check(eng.lib.jn_detector_step(eng.handle, ptr(x[i:i + n]), n, ptr(t[i:i + n]), t.shape[1],
                                           float(loss_scale) * n / N, ptr(metrics), _lib.current_stream(g.device)),
                  "jn_detector_step")
            for j, k in enumerate(LOSS_NAMES):
                tot[k] = tot[k] + metrics[j] * (n / N)
        if getattr(g, "_flat_grads", None) is not None and getattr(g, "_publish_detector_grads", True):
            g.publish_engine_grads()                   # bound model (autograd bridge): yolox.*.grad follow at once
        return tot

    @staticmethod
    def clamp_outputs(outputs, image_size: int):
        """src/models/yolox.py:93-113."""
        for b in outputs:
            if b is not None:
                b[:, :4].clamp_(min=0, max=image_size - 1)
        return outputs
