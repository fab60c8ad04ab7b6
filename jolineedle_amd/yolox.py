"""``NeedleYOLOX`` — detector wrapper of the reference (src/models/yolox.py:15-120): inference branch
(``jn_detect``) and the training loss branch (``jn_detector_step``), computed by libjnroll.so."""
from typing import List, Optional

import torch

from . import _lib
from ._lib import check, ptr


class NeedleYOLOX:
    """View on the ``yolox.*`` part of a GPT's engine.

    ``forward(patches, targets=None) -> (outputs, fpn_outs, losses)`` as src/models/yolox.py:24-91.
    With ``targets`` ([N, nb, 5] = class, x1, y1, x2, y2; zero rows = padding) the engine runs the detector in
    train mode (batch-statistics BN), the SimOTA assignment and the IoU / objectness / class / L1 losses AND the
    backward pass in one call: the yolox.* gradients are accumulated in the engine's gradient arena (what
    ``total_loss.backward()`` does in the reference, src/reinforce.py:336-341), scaled by ``loss_scale``.
    Deviation: in that case ``outputs`` / ``fpn_outs`` are not produced (the reference's training loop discards
    them); call ``forward(patches)`` for predictions.
    """

    def __init__(self, gpt, conf_threshold: float):
        self._gpt = gpt
        self.conf_threshold = conf_threshold

    def __call__(self, patches, targets=None, loss_scale: float = 1.0):
        return self.forward(patches, targets, loss_scale)

    def forward(self, patches: torch.Tensor, targets: Optional[torch.Tensor] = None, loss_scale: float = 1.0):
        if targets is not None:
            return [None] * patches.shape[0], None, self.loss_and_backward(patches, targets, loss_scale)
        g = self._gpt
        g.sync_weights()
        eng = g.engine()
        N, P = patches.shape[0], g.patch_size
        K = eng.cfg.max_det_per_patch
        x = patches.to(g.device, torch.float32).contiguous()
        boxes = torch.zeros((N, K, 7), device=g.device)
        counts = torch.zeros((N,), device=g.device, dtype=torch.int32)
        for i in range(0, N, g.max_batch):
            n = min(g.max_batch, N - i)
            check(eng.lib.jn_detect(eng.handle, ptr(x[i:i + n]), n, ptr(boxes[i:i + n]), ptr(counts[i:i + n]),
                                    None, _lib.current_stream(g.device)), "jn_detect")
        fpn_outs = g.backbone_features(x, _lib.JN_NET_DETECTOR)
        cnt = counts.tolist()
        outputs: List[Optional[torch.Tensor]] = [boxes[i, :c].clone() if c > 0 else None for i, c in enumerate(cnt)]
        return outputs, fpn_outs, {}

    def loss_and_backward(self, patches: torch.Tensor, targets: torch.Tensor, loss_scale: float = 1.0) -> dict:
        """Loss branch of src/models/yolox.py:58-73 + backward; returns the reference's loss dict (device scalars)."""
        g = self._gpt
        g.sync_weights()
        eng = g.engine()
        N = patches.shape[0]
        assert targets.shape[0] == N and targets.shape[-1] >= 5
        x = patches.to(g.device, torch.float32).contiguous()
        t = targets[..., :5].to(g.device, torch.float32).contiguous()
        names = ("total_loss", "iou_loss", "conf_loss", "cls_loss", "l1_loss", "num_fg")
        tot = {k: torch.zeros((), device=g.device) for k in names}
        # The reference feeds the whole detection batch at once (BN statistics and the 1 / num_fg normalisation span
        # all patches).  Up to max_batch patches that is what happens here; a larger batch is processed in chunks of
        # max_batch, each weighted by its share of the patches (deviation: per-chunk statistics / normalisation).
        for i in range(0, N, g.max_batch):
            n = min(g.max_batch, N - i)
            metrics = torch.zeros(8, device=g.device)
            check(eng.lib.jn_detector_step(eng.handle, ptr(x[i:i + n]), n, ptr(t[i:i + n]), t.shape[1],
                                           float(loss_scale) * n / N, ptr(metrics), _lib.current_stream(g.device)),
                  "jn_detector_step")
            for j, k in enumerate(names):
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
