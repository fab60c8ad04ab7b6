"""Oracle restatement of the batched needle environment.

TEST INFRASTRUCTURE ONLY.  Follows src/env/common.py:4-56 (action vocabulary) and
src/env/general_env.py:117-379 (state, reset, step, rewards, terminated,
patch gather, bbox -> patch-grid masks) plus :381-504, 548-573 (per-patch
target splitting).  Pinned by tests/golden/make_golden.py (G1, G7 vectors).

kornia ``Boxes.from_tensor(..., "xyxy_plus").to_mask`` (general_env.py:373-374)
is third-party and absent: **parity unpinned** for the inclusive-max rule
(a box covers pixels x1..x2 and y1..y2 inclusive, clipped to the image), which
is restated analytically on the patch grid instead of through a pixel mask.
"""
from typing import List, Optional, Tuple

import torch

# src/env/common.py:4-27
ACTION_NAMES = ["LEFT", "RIGHT", "UP", "DOWN", "LEFT_UP", "RIGHT_UP", "LEFT_DOWN", "RIGHT_DOWN", "STOP"]
ACTION_DELTAS_YX = [(0, -1), (0, 1), (-1, 0), (1, 0), (-1, -1), (-1, 1), (1, -1), (1, 1), (0, 0)]
STOP = 8


def n_action_classes(stop_enabled: bool) -> int:
    """get_actions_info, src/env/common.py:48-56."""
    return 9 if stop_enabled else 8


def bboxes_to_grid_masks(bboxes: torch.Tensor, height: int, width: int, patch: int) -> torch.Tensor:
    """[B, nb, 4] int xyxy (inclusive max) -> [B, Gh, Gw] bool: patch holds >=1 box pixel.
    general_env.py:360-379 without materialising the [B, nb, H, W] pixel mask."""
    B = bboxes.shape[0]
    gh, gw = height // patch, width // patch
    out = torch.zeros((B, gh, gw), dtype=torch.bool)
    for b in range(B):
        for x1, y1, x2, y2 in bboxes[b].tolist():
            x1, y1 = max(int(x1), 0), max(int(y1), 0)
            x2, y2 = min(int(x2) + 1, width), min(int(y2) + 1, height)   # exclusive
            if x2 <= x1 or y2 <= y1:
                continue
            out[b, y1 // patch:(y2 - 1) // patch + 1, x1 // patch:(x2 - 1) // patch + 1] = True
    return out


class EnvRef:
    """Functional twin of NeedleGeneralEnv for n_glimps_levels == 1 (the only value
    the REINFORCE trainer uses, src/reinforce.py:58)."""

    def __init__(self, images: torch.Tensor, bboxes: torch.Tensor, patch_size: int,
                 max_ep_len: int, n_glimps_levels: int = 1, stop_enabled: bool = False):
        if not torch.is_tensor(images):               # shape tuple: state-only env (no patch gather)
            self._shape_only = tuple(images)
            images = torch.zeros((images[0], images[1], 1, 1)).expand(*images)
        assert images.shape[0] == bboxes.shape[0] and images.dim() == 4
        assert n_glimps_levels == 1
        self.patch_size, self.max_ep_len, self.stop_enabled = patch_size, max_ep_len, stop_enabled
        self.batch_size, self.n_channels, self.height, self.width = images.shape
        assert self.height % patch_size == 0 and self.width % patch_size == 0
        self.n_vertical_patches = self.height // patch_size
        self.n_horizontal_patches = self.width // patch_size
        self.images = images.unsqueeze(1)           # [B, 1, C, H, W] (general_env.py:115)
        self.bboxes = bboxes
        self.bbox_masks = bboxes_to_grid_masks(bboxes, self.height, self.width, patch_size)
        self._zero_state()

    def _zero_state(self):
        B = self.batch_size
        self.positions = torch.zeros((B, 2), dtype=torch.long)
        self.visited_patches = torch.zeros_like(self.bbox_masks)
        self.steps = torch.zeros((B,), dtype=torch.long)
        self.has_stopped = torch.zeros((B,), dtype=torch.bool)

    def _mark_visited(self):
        idx = torch.arange(self.batch_size)
        self.visited_patches = self.visited_patches.clone()
        self.visited_patches[idx, self.positions[:, 0], self.positions[:, 1]] = True

    @property
    def patches(self) -> torch.Tensor:
        if getattr(self, "_shape_only", None):
            return None
        P = self.patch_size
        return torch.stack([
            self.images[i, :, :, int(y) * P:(int(y) + 1) * P, int(x) * P:(int(x) + 1) * P]
            for i, (y, x) in enumerate(self.positions.tolist())])

    def reset(self, positions: Optional[torch.Tensor] = None):
        self._zero_state()
        if positions is not None:
            self.positions = positions.clone()
        else:                                        # CPU default generator, general_env.py:158-163
            self.positions[:, 0] = torch.randint(0, self.n_vertical_patches, (self.batch_size,))
            self.positions[:, 1] = torch.randint(0, self.n_horizontal_patches, (self.batch_size,))
        self._mark_visited()
        return self.patches, {"positions": self.positions}

    @property
    def terminated(self) -> torch.Tensor:
        if self.stop_enabled:
            return self.has_stopped
        return ((self.bbox_masks & self.visited_patches) != self.bbox_masks).sum(dim=(1, 2)) == 0

    def _rewards(self) -> torch.Tensor:
        """general_env.py:321-358; uses ``visited`` from BEFORE this step's update."""
        idx = torch.arange(self.batch_size)
        y, x = self.positions[:, 0], self.positions[:, 1]
        hit = self.bbox_masks[idx, y, x] & ~self.visited_patches[idx, y, x]
        cost = torch.ones_like(hit) * (-1 / self.max_ep_len)     # bool * float -> float32
        stop_eval = torch.zeros_like(hit)
        if self.stop_enabled:
            found = (self.visited_patches & self.bbox_masks).sum(dim=(1, 2))
            total = self.bbox_masks.sum(dim=(1, 2))
            all_found = (found == total).to(torch.int64)
            stop_eval = (all_found * found + (1 - all_found) * (found - total)) * self.has_stopped
        return hit + cost + stop_eval

    @torch.no_grad()
    def step(self, actions: torch.Tensor):
        deltas = torch.tensor([ACTION_DELTAS_YX[int(a)] for a in actions.tolist()])
        self.positions = self.positions + deltas
        self.positions[:, 0].clamp_(0, self.n_vertical_patches - 1)
        self.positions[:, 1].clamp_(0, self.n_horizontal_patches - 1)
        self.has_stopped = self.has_stopped | (actions == STOP)
        rewards = self._rewards()
        self._mark_visited()
        self.steps = self.steps + 1
        truncated = self.steps >= self.max_ep_len
        return self.patches, rewards, self.terminated, truncated, {"positions": self.positions}

    @property
    def prop_patches_found(self) -> torch.Tensor:
        count = (self.bbox_masks & self.visited_patches).sum(dim=(1, 2))
        tot = self.bbox_masks.sum(dim=(1, 2))
        tot[tot == 0] = 1
        return count / tot

    @property
    def prop_bboxes_found(self) -> torch.Tensor:
        return (self.prop_patches_found > 0).to(torch.float32)

    # ---- per-patch target splitting (general_env.py:381-504, 548-573) --------------
    def split_bboxes(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """[B, Gh, Gw, nb, 4] patch-local xyxy + [B, Gh, Gw, nb] validity."""
        P = self.patch_size
        nb = self.bboxes.shape[1]
        out = torch.zeros((self.batch_size, self.n_vertical_patches, self.n_horizontal_patches, nb, 4),
                          dtype=torch.long)
        valid = torch.zeros(out.shape[:-1], dtype=torch.bool)
        for b in range(self.batch_size):
            for k, (x1, y1, x2, y2) in enumerate(self.bboxes[b].int().tolist()):
                for gy in range(y1 // P, y2 // P + 1):
                    for gx in range(x1 // P, x2 // P + 1):
                        lx1 = max(x1, gx * P) - gx * P
                        ly1 = max(y1, gy * P) - gy * P
                        lx2 = min(x2, gx * P + P - 1) - gx * P
                        ly2 = min(y2, gy * P + P - 1) - gy * P
                        out[b, gy, gx, k] = torch.tensor([lx1, ly1, lx2, ly2])
                        valid[b, gy, gx, k] = True
        return out, valid

    def get_detection_targets(self) -> List[torch.Tensor]:
        boxes, _ = self.split_bboxes()
        P = self.patch_size
        res = []
        for b in range(self.batch_size):
            rows = []
            for gy in range(boxes.shape[1]):
                for gx in range(boxes.shape[2]):
                    for k in range(boxes.shape[3]):
                        box = boxes[b, gy, gx, k]
                        if int(box.abs().sum()) == 0:
                            continue
                        off = torch.tensor([gx * P, gy * P, gx * P, gy * P])
                        rows.append(torch.cat((torch.zeros(1, dtype=torch.long), box + off)))
            res.append(torch.stack(rows))
        return res
