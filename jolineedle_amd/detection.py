"""Detection bookkeeping around the rollout: splitting ground-truth boxes over the patch grid,
patch -> full-image box coordinates, merging of contiguous boxes and mAP-50.  Host-side integer / float logic on small
tensors (a few boxes per image) — nothing here is on the per-glimpse hot path.

Reference: ``NeedleGeneralEnv.parse_bboxes / get_detection_targets`` (src/env/general_env.py:381-573),
``Trainer.patch_bboxes2full_image`` (src/trainer.py:250-280), ``merge_boxes`` (src/utils.py:185-255),
``Trainer.compute_detection_metrics`` (src/trainer.py:188-248; the reference calls torchmetrics' COCO
MeanAveragePrecision, which is not vendored — map_50 is restated here from the published COCO protocol and pinned by
the known answers of the reference's tests/test_map.py: 0, 1 and 0.8)."""
from typing import List, Optional, Tuple

import torch
from torch import Tensor


def split_bboxes_over_patches(bboxes: Tensor, n_vertical: int, n_horizontal: int, patch_size: int) -> Tuple[Tensor, Tensor]:
    """[B, nb, 4] xyxy image boxes -> per-patch local boxes [B, Gy, Gx, nb, 4] + masks [B, Gy, Gx, nb].

    A box that crosses a patch border is cut at the border (inclusive max = patch_size - 1) and continues in the
    neighbouring patch(es), as the reference's recursive placement does (general_env.py:432-490).  Zero-padded rows
    land in patch (0, 0) with an all-zero box (a reference quirk that its callers filter by |box| == 0)."""
    P = patch_size
    bb = bboxes.to(torch.int64).cpu()
    B, nb = bb.shape[0], bb.shape[1]
    out = torch.zeros((B, n_vertical, n_horizontal, nb, 4), dtype=torch.long)
    masks = torch.zeros((B, n_vertical, n_horizontal, nb), dtype=torch.bool)
    for b in range(B):
        for k in range(nb):
            x1, y1, x2, y2 = (int(v) for v in bb[b, k])
            # the recursion of the reference visits exactly the grid cells the box touches; cell p spans
            # [p * P, p * P + P - 1] (inclusive) and the next piece starts at (p + 1) * P
            for py in range(y1 // P, y2 // P + 1):
                for px in range(x1 // P, x2 // P + 1):
                    if not (0 <= py < n_vertical and 0 <= px < n_horizontal):
                        continue
                    out[b, py, px, k] = torch.tensor([max(x1, px * P) - px * P, max(y1, py * P) - py * P,
                                                      min(x2, px * P + P - 1) - px * P, min(y2, py * P + P - 1) - py * P])
                    masks[b, py, px, k] = True
    return out.to(bboxes.device), masks.to(bboxes.device)


def detection_targets(bboxes: Tensor, n_vertical: int, n_horizontal: int, patch_size: int) -> List[Tensor]:
    """Full-image targets [n, 5] = (class 0, x1, y1, x2, y2), one entry per (box, patch) piece, in (y, x, box) order
    (general_env.py:546-573)."""
    local, _ = split_bboxes_over_patches(bboxes, n_vertical, n_horizontal, patch_size)
    res = []
    for b in range(local.shape[0]):
        rows = []
        for y in range(n_vertical):
            for x in range(n_horizontal):
                for k in range(local.shape[3]):
                    box = local[b, y, x, k]
                    if int(box.abs().sum()) == 0:
                        continue
                    off = torch.tensor([x, y, x, y], device=box.device) * patch_size
                    rows.append(torch.cat((torch.zeros(1, dtype=box.dtype, device=box.device), box + off)))
        res.append(torch.stack(rows) if rows else torch.zeros((0, 5), dtype=torch.long, device=bboxes.device))
    return res


def patch_bboxes2full_image(outputs: List[List[Optional[Tensor]]], offsets: Tensor,
                            masks: Optional[Tensor] = None) -> List[Optional[Tensor]]:
    """Per-patch predictions (list over images of lists over glimpse steps) -> one tensor of boxes per image in
    full-image coordinates; offsets[i, j] = (x, y) of patch j of image i (src/trainer.py:250-280)."""
    res = []
    for i, per_image in enumerate(outputs):
        kept = []
        for j, boxes in enumerate(per_image):
            if masks is not None and not bool(masks[i, j]):
                continue
            if boxes is None:
                continue
            moved = boxes.clone()
            moved[:, 0:2] += offsets[i, j].to(moved.dtype)
            moved[:, 2:4] += offsets[i, j].to(moved.dtype)
            kept.append(moved)
        res.append(torch.cat(kept) if kept else None)
    return res


def merge_boxes(boxes: Tensor, threshold: int = 2, target: bool = False) -> Tensor:
    """Union of boxes whose edges are within `threshold` px of each other (src/utils.py:198-255): box i opens a group
    (or keeps the one it already belongs to) and pulls in every later box j with min edge distance <= threshold.
    Predictions (x1, y1, x2, y2, obj, cls, ...) keep the best obj * cls of the group; targets are (cls, x1, y1, x2, y2)."""
    off = 1 if target else 0
    n = len(boxes)
    group_of = [-1] * n
    groups: List[List[int]] = []
    for i in range(n):
        if group_of[i] < 0:
            group_of[i] = len(groups)
            groups.append([i])
        gi = group_of[i]
        a = boxes[i]
        for j in range(i + 1, n):
            b = boxes[j]
            d = min(abs(float(b[off + 2] - a[off + 0])), abs(float(a[off + 2] - b[off + 0])),
                    abs(float(b[off + 3] - a[off + 1])), abs(float(a[off + 3] - b[off + 1])))
            if d <= threshold:
                groups[gi].append(j)             # (the reference appends duplicates too; min / max ignore them)
                if group_of[j] < 0:
                    group_of[j] = gi
    merged = []
    for grp in groups:
        sel = boxes[sorted(set(grp))]
        row = [sel[:, off + 0].min(), sel[:, off + 1].min(), sel[:, off + 2].max(), sel[:, off + 3].max()]
        if target:
            row = [torch.zeros((), dtype=boxes.dtype, device=boxes.device)] + row
        elif boxes.shape[1] > 5:
            row += [(sel[:, 4] * sel[:, 5]).max(), torch.ones((), dtype=boxes.dtype, device=boxes.device)]
        merged.append(torch.stack([torch.as_tensor(v, dtype=boxes.dtype, device=boxes.device) for v in row]))
    return torch.stack(merged)


def merge_boxes_batched(batch: List[Optional[Tensor]], threshold: int = 2, target: bool = False) -> List[Optional[Tensor]]:
    return [None if b is None else merge_boxes(b, threshold, target) for b in batch]


def _iou_matrix(a: Tensor, b: Tensor) -> Tensor:
    lt = torch.maximum(a[:, None, :2], b[None, :, :2])
    rb = torch.minimum(a[:, None, 2:4], b[None, :, 2:4])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (area_a[:, None] + area_b[None, :] - inter).clamp(min=1e-12)


def map_50(outputs: List[Optional[Tensor]], targets: List[Tensor], max_det: int = 100) -> float:
    """COCO-protocol AP at IoU 0.5 for the single needle class: per image the `max_det` best predictions are matched
    greedily in score order to the not-yet-matched target of highest IoU (>= 0.5); precision is made monotone and
    sampled at the 101 recall points 0, 0.01, ..., 1."""
    n_gt = sum(len(t) for t in targets)
    if n_gt == 0:
        return 0.0                               # src/trainer.py:205-208
    scores, hits = [], []
    for out, tgt in zip(outputs, targets):
        if out is None or len(out) == 0:
            continue
        out = out.detach().to("cpu", torch.float64)              # host-side bookkeeping on a handful of boxes
        order = torch.argsort(out[:, 4], descending=True, stable=True)[:max_det]
        boxes, sc = out[order, :4], out[order, 4]
        gt = tgt[:, 1:5].detach().to("cpu", torch.float64)
        taken = torch.zeros(len(gt), dtype=torch.bool)
        iou = _iou_matrix(boxes, gt) if len(gt) else torch.zeros((len(boxes), 0), dtype=torch.float64)
        for p in range(len(boxes)):
            best, best_j = 0.5, -1
            for j in range(len(gt)):
                if taken[j]:
                    continue
                if iou[p, j] >= best:
                    best, best_j = float(iou[p, j]), j
            if best_j >= 0:
                taken[best_j] = True
            scores.append(float(sc[p]))
            hits.append(best_j >= 0)
    if not scores:
        return 0.0
    order = sorted(range(len(scores)), key=lambda i: -scores[i])
    tp = torch.tensor([1.0 if hits[i] else 0.0 for i in order], dtype=torch.float64).cumsum(0)
    fp = torch.tensor([0.0 if hits[i] else 1.0 for i in order], dtype=torch.float64).cumsum(0)
    recall = tp / n_gt
    precision = tp / (tp + fp)
    for i in range(len(precision) - 2, -1, -1):
        precision[i] = max(precision[i], precision[i + 1])
    ap = 0.0
    for r in torch.linspace(0, 1, 101, dtype=torch.float64):
        idx = int(torch.searchsorted(recall, r, right=False))
        ap += float(precision[idx]) if idx < len(precision) else 0.0
    return ap / 101.0


def compute_detection_metrics(outputs: List[Optional[Tensor]], targets: List[Tensor]) -> dict:
    """``Trainer.compute_detection_metrics`` (src/trainer.py:188-248): {"map": mAP-50 over the batch}."""
    dev = targets[0].device if len(targets) else torch.device("cpu")
    return {"map": torch.tensor([map_50(outputs, targets)], dtype=torch.float32, device=dev)}
