"""``infer.infer`` of the reference (infer.py:82-221) on the accelerated rollout, without the plotting: per image — scale
to [0, 1], zero-pad to a multiple of the patch size (bottom / right), one sampled (or greedy) rollout with the detector
on every visited patch, boxes moved to full-image coordinates; with targets also the rollout and detection metrics."""
import time
from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from .detection import compute_detection_metrics, patch_bboxes2full_image
from .env import NeedleGeneralEnv


def load_bboxes(bbox_fname) -> List[List[int]]:
    """One ``cls x1 y1 x2 y2`` line per box (infer.py:71-79)."""
    out = []
    with open(bbox_fname) as f:
        for line in f:
            parts = line.strip().split()
            if len(parts) >= 5:
                out.append([int(v) for v in parts[1:5]])
    return out


def pad_to_patch_multiple(image: torch.Tensor, patch_size: int) -> torch.Tensor:
    """[1, C, H, W] -> zero-padded bottom / right so that H and W are multiples of `patch_size` (infer.py:138-146)."""
    H, W = image.shape[-2:]
    ph = ((H - 1) // patch_size + 1) * patch_size
    pw = ((W - 1) // patch_size + 1) * patch_size
    return F.pad(image, (0, pw - W, 0, ph - H), value=0)


@torch.no_grad()
def infer_images(trainer, images: Sequence[torch.Tensor], targets: Optional[Sequence] = None, sample_actions: bool = True,
                 do_detection: Optional[bool] = None) -> Dict:
    """`images`: [C, H, W] tensors (uint8 0..255 or float 0..1) of any sizes; `targets`: per image an [n, 4] xyxy list /
    tensor or None.  Returns per-image boxes ([n, 7] in full-image pixels or None), positions, step counts, durations
    and — where targets are given — the mean of the reference's metrics."""
    cfg, dev = trainer.config, trainer.device
    P, T = int(cfg.patch_size), int(cfg.max_seq_len)
    if do_detection is None:
        do_detection = bool(getattr(cfg, "detection_enabled", False)) and trainer.yolox_model() is not None
    res = {"boxes": [], "positions": [], "steps": [], "duration_ms": []}
    all_metrics = defaultdict(list)
    for i, img in enumerate(images):
        x = img.to(dev)
        x = (x.float() / 255 if not x.is_floating_point() else x.float()).unsqueeze(0)
        x = pad_to_patch_multiple(x, P).contiguous()
        tg = None if targets is None or i >= len(targets) or targets[i] is None else torch.as_tensor(targets[i]).reshape(1, -1, 4)
        bboxes = tg.to(torch.long) if tg is not None else torch.zeros((1, 1, 4), dtype=torch.long)
        env = NeedleGeneralEnv(x, bboxes, P, T, 1, bool(getattr(cfg, "stop_enabled", False)))
        t0 = time.perf_counter()
        ro = trainer.rollout(env, do_detection=do_detection, sample_actions=sample_actions)
        torch.cuda.synchronize(dev)
        res["duration_ms"].append((time.perf_counter() - t0) * 1e3)
        offsets = ro["positions"][:, :, [1, 0]] * P                       # (y, x) grid -> (x, y) pixels
        full = patch_bboxes2full_image(ro["bboxes"], offsets, ro["masks"])
        res["boxes"].append(full[0])
        res["positions"].append(ro["positions"][0][ro["masks"][0]].cpu())
        res["steps"].append(int(ro["rewards"].shape[1]))
        if tg is not None:
            m = trainer.compute_metrics(ro, env)
            if do_detection:
                m.update(compute_detection_metrics(full, env.get_detection_targets()))
            for k, v in m.items():
                all_metrics[k].append(float(v))
    res["metrics"] = {k: sum(v) / len(v) for k, v in all_metrics.items()}
    return res
