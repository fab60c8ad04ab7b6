"""``NeedleYOLOX`` — detector wrapper of the reference (src/models/yolox.py:15-120),
inference branch, computed by libjnroll.so (``jn_detect``)."""
from typing import List, Optional

import torch

from . import _lib
from ._lib import check, ptr


class NeedleYOLOX:
    """View on the ``yolox.*`` part of a GPT's engine.

    ``forward(patches, targets=None) -> (outputs, fpn_outs, losses)`` as
    src/models/yolox.py:24-91; ``targets`` (the SimOTA loss branch, :58-73) is a "next"
    row of the scope table and raises NotImplementedError.
    """

    def __init__(self, gpt, conf_threshold: float):
        self._gpt = gpt
        self.conf_threshold = conf_threshold

    def __call__(self, patches, targets=None):
        return self.forward(patches, targets)

    def forward(self, patches: torch.Tensor, targets: Optional[torch.Tensor] = None):
        if targets is not None:
            raise NotImplementedError("detector loss branch (src/models/yolox.py:58-73) is not part of this build")
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

    @staticmethod
    def clamp_outputs(outputs, image_size: int):
        """src/models/yolox.py:93-113."""
        for b in outputs:
            if b is not None:
                b[:, :4].clamp_(min=0, max=image_size - 1)
        return outputs
