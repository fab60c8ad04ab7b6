"""Batch assembly feeding the env: the reference's padded collate layout and a device-side
synthetic batch of that layout (no 241 MB/image host-to-device copy; there are no datasets on the target box).

Reference: ``NeedleDataset.padded_collate_fn`` (src/dataset.py:307-347)."""
from typing import Dict, List, Sequence

import torch
from torch import Tensor


def padded_collate(images: Sequence[Tensor], bboxes: Sequence[Tensor], patch_size: int, class_ids: Sequence[int] = None) -> Dict[str, Tensor]:
    """Images [3, H_i, W_i] of varying sizes are zero-padded (bottom / right) to the largest one, rounded up to a
    multiple of `patch_size`; boxes [n_i, 4] (xyxy pixels) are zero-row padded to the longest list."""
    assert len(images) == len(bboxes) and len(images) > 0
    max_h = max(int(im.shape[1]) for im in images)
    max_w = max(int(im.shape[2]) for im in images)
    max_nb = max(int(b.shape[0]) for b in bboxes)
    H = -(-max_h // patch_size) * patch_size
    W = -(-max_w // patch_size) * patch_size
    dev, dt = images[0].device, images[0].dtype
    out = torch.zeros((len(images), images[0].shape[0], H, W), dtype=dt, device=dev)
    bb = torch.zeros((len(images), max_nb, 4), dtype=torch.long, device=bboxes[0].device)
    for i, (im, b) in enumerate(zip(images, bboxes)):
        out[i, :, :im.shape[1], :im.shape[2]] = im
        if b.shape[0]:
            bb[i, :b.shape[0]] = b.to(torch.long)
    res = {"image": out, "bboxes": bb}
    res["class_id"] = torch.tensor(list(class_ids) if class_ids is not None else [0] * len(images))
    return res


def synthetic_batch(batch_size: int, grid: int, patch_size: int, seed: int, device="cuda", max_boxes: int = 3) -> Dict[str, Tensor]:
    """SURVEY.md §8(d) synthetic inputs, generated ON the device: images uniform in [0, 1) of (grid * patch_size)^2
    pixels, 1..max_boxes boxes per image with sides in [32, patch_size) fully inside the image (int64 xyxy, zero-row
    padded), start positions uniform over the grid.  Deterministic in `seed`."""
    side = grid * patch_size
    gen = torch.Generator(device=device).manual_seed(seed)
    images = torch.rand((batch_size, 3, side, side), device=device, generator=gen)
    g = torch.Generator().manual_seed(seed)
    boxes = torch.zeros((batch_size, max_boxes, 4), dtype=torch.long)
    for b in range(batch_size):
        for k in range(int(torch.randint(1, max_boxes + 1, (1,), generator=g))):
            w, h = (int(torch.randint(32, patch_size, (1,), generator=g)) for _ in range(2))
            x = int(torch.randint(0, side - w, (1,), generator=g))
            y = int(torch.randint(0, side - h, (1,), generator=g))
            boxes[b, k] = torch.tensor([x, y, x + w, y + h])
    start = torch.randint(0, grid, (batch_size, 2), generator=g)
    return {"image": images, "bboxes": boxes, "class_id": torch.zeros(batch_size, dtype=torch.long), "start_positions": start}
