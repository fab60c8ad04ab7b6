"""Teacher trajectories for the supervised step: ``NeedleSimpleEnv`` of the reference
(src/env/simple_env.py:166-764) re-designed as an INDEX generator + one device gather.

The reference walks one image on the host, slicing a [C, P, P] patch out of a CPU image at every step and stacking
them (a 241 MB image per sample crosses PCIe first).  Here the walk produces only integers — grid positions,
actions taken, teacher actions, labels, local boxes — and the patches of a whole batch of trajectories are gathered
from the device-resident images by ONE ``jn_gather_patches_indexed`` launch (bit-exact strided copy; masked steps
are zero patches exactly like the reference's zero-initialised sample).

Random streams are consumed in the reference's order (``numpy.random.default_rng(seed)`` for positions / keypoints /
STOP replacement, Python's ``random`` for nearest-neighbour ties), so a seeded walk reproduces the reference's —
tests/golden/g8_trajectories.npz holds walks recorded from the reference itself.

Positions are (y, x) tuples on the patch grid; boxes are pixel (x1, y1, x2, y2).
"""
import random as _random
from itertools import product
from typing import Dict, List, Optional, Sequence, Set, Tuple

import numpy as np
import torch

from .common import ACTION_DELTAS, MOVES, Action

Pos = Tuple[int, int]

_BY_SIGN = {  # (sign dy, sign dx) -> action       (move_towards, src/env/simple_env.py:84-125)
    (1, 0): Action.DOWN, (-1, 0): Action.UP, (0, 1): Action.RIGHT, (0, -1): Action.LEFT,
    (-1, 1): Action.RIGHT_UP, (-1, -1): Action.LEFT_UP, (1, 1): Action.RIGHT_DOWN, (1, -1): Action.LEFT_DOWN,
    (0, 0): Action.STOP,
}


def _sign(v: int) -> int:
    return (v > 0) - (v < 0)


def move_towards(current: Pos, target: Pos) -> Action:
    """One king's-move step from `current` towards `target` (STOP when already there)."""
    return _BY_SIGN[(_sign(target[0] - current[0]), _sign(target[1] - current[1]))]


def _boxes_xyxy(bboxes) -> List[Tuple[int, int, int, int]]:
    """Accepts an [n, 4] xyxy tensor / array or the reference's list of BBox(up_left=(y, x), bottom_right=(y, x))."""
    if isinstance(bboxes, (torch.Tensor, np.ndarray)):
        return [tuple(int(v) for v in row) for row in np.asarray(bboxes.cpu() if isinstance(bboxes, torch.Tensor) else bboxes).reshape(-1, 4)]
    out = []
    for b in bboxes:
        (y1, x1), (y2, x2) = b
        out.append((int(x1), int(y1), int(x2), int(y2)))
    return out


class NeedleSimpleEnv:
    """Single-image teacher.  `image` may be None (index generation only), a [C, H, W] tensor, or — to share one
    device batch between many envs — a ([B, C, H, W] tensor, image index) pair."""

    def __init__(self, image, patch_size: int, bboxes, seed: Optional[int] = None, height: Optional[int] = None,
                 width: Optional[int] = None, py_random=None):
        self.patch_size = int(patch_size)
        self.rng = np.random.default_rng(seed)
        self.py_random = py_random if py_random is not None else _random
        self.raw_bboxes = _boxes_xyxy(bboxes)
        self.image, self.image_index = None, 0
        if image is not None:
            if isinstance(image, tuple):
                self.image, self.image_index = image[0], int(image[1])
            else:
                self.image = image[None]
            self.n_channels, self.height, self.width = (int(s) for s in self.image.shape[1:])
        else:
            assert height is not None and width is not None, "image or (height, width) required"
            self.n_channels, self.height, self.width = 3, int(height), int(width)
        P = self.patch_size
        self.patch_height, self.patch_width = self.height // P, self.width // P
        # boxes on the patch grid, (y1, x1, y2, x2) inclusive
        self.bboxes = [(y1 // P, x1 // P, y2 // P, x2 // P) for x1, y1, x2, y2 in self.raw_bboxes]
        self.position: Pos = (0, 0)
        self.bbox_patches: Set[Pos] = set()
        for box in self.raw_bboxes:
            self.bbox_patches = self.bbox_patches | self.bbox_positions(box)
        self.visited_bbox_patches: Set[Pos] = set()

    # ---- geometry -------------------------------------------------------------------------------------
    def bbox_positions(self, box, area_threshold: float = 0.05) -> Set[Pos]:
        """Grid cells holding more than `area_threshold` of a patch area of the box, plus the cell of the box centre,
        restricted to the grid (src/env/simple_env.py:270-321)."""
        P = self.patch_size
        x1, y1, x2, y2 = box
        cells: Set[Pos] = set()
        for y, x in product(range(y1 // P, y2 // P + 1), range(x1 // P, x2 // P + 1)):
            h = min((y + 1) * P, y2) - max(y * P, y1)
            w = min((x + 1) * P, x2) - max(x * P, x1)
            if h * w / (P ** 2) > area_threshold:
                cells.add((y, x))
        cells.add((((y1 + y2) // 2) // P, ((x1 + x2) // 2) // P))
        cells = {c for c in cells if 0 <= c[1] < self.patch_width}
        cells = {c for c in cells if 0 <= c[0] < self.patch_height}
        return cells

    def local_bboxes(self, position: Optional[Pos] = None) -> np.ndarray:
        """[n_boxes, 6] rows (class 0, x1, y1, x2, y2 relative to the patch, objectness 1) of the parts of the raw
        boxes inside the patch at `position`; zero rows where there is no overlap (src/env/simple_env.py:231-268)."""
        y, x = self.position if position is None else position
        P = self.patch_size
        px, py = x * P, y * P
        out = np.zeros((len(self.raw_bboxes), 6), dtype=np.float32)
        for i, (x1, y1, x2, y2) in enumerate(self.raw_bboxes):
            cx1, cy1, cx2, cy2 = max(px, x1), max(py, y1), min(px + P, x2), min(py + P, y2)
            if cx1 < cx2 and cy1 < cy2:
                out[i] = (0, cx1 - px, cy1 - py, cx2 - px, cy2 - py, 1)
        return out

    # ---- walking --------------------------------------------------------------------------------------
    def reset(self, position: Optional[Pos] = None, visited_bbox_patches: Optional[Set[Pos]] = None) -> Dict:
        if position is None:
            y = int(self.rng.integers(low=0, high=self.patch_height))
            x = int(self.rng.integers(low=0, high=self.patch_width))
            position = (y, x)
        self.position = (int(position[0]), int(position[1]))
        self.visited_bbox_patches = set() if visited_bbox_patches is None else visited_bbox_patches
        if self.position in self.bbox_patches:
            self.visited_bbox_patches.add(self.position)
        return self.gather_infos()

    def step(self, move) -> Dict:
        move = move if isinstance(move, Action) else Action(int(move))
        dy, dx = ACTION_DELTAS[move]
        self.position = (min(max(self.position[0] + dy, 0), self.patch_height - 1),
                         min(max(self.position[1] + dx, 0), self.patch_width - 1))
        if self.position in self.bbox_patches:
            self.visited_bbox_patches.add(self.position)
        return self.gather_infos()

    def gather_infos(self) -> Dict:
        return {"position": self.position, "number_patches_found": len(self.visited_bbox_patches),
                "local_bboxes": self.local_bboxes(), "inside_bbox": self.position in self.bbox_patches}

    def remove_stop_action(self, action: Action) -> Action:
        return MOVES[int(self.rng.choice(len(MOVES)))] if action is Action.STOP else action

    def generate_keypoints(self, n: int) -> List[Pos]:
        pts = []
        for _ in range(n):
            y = int(self.rng.integers(0, self.patch_height))
            x = int(self.rng.integers(0, self.patch_width))
            pts.append((y, x))
        return pts

    def generate_binomial_keypoints(self, n: int, target: Pos) -> List[Pos]:
        """Binomial(grid, 1/2) displacement around `target`, wrapping on the grid (src/env/simple_env.py:684-713)."""
        pts = []
        for _ in range(n):
            dx = int(self.rng.binomial(self.patch_width, 0.5)) - self.patch_width // 2
            dy = int(self.rng.binomial(self.patch_height, 0.5)) - self.patch_height // 2
            pts.append(((target[0] + dy) % self.patch_height, (target[1] + dx) % self.patch_width))
        return pts

    def build_keypoints_trajectory(self) -> List[Pos]:
        """Greedy nearest-neighbour (L1) order over the not-yet-visited box cells, ties broken by Python's `random`;
        one random cell when there is nothing to visit (src/env/simple_env.py:590-629)."""
        todo: Set[Pos] = set()
        for box in self.raw_bboxes:
            todo |= self.bbox_positions(box)
        for p in self.visited_bbox_patches:
            todo.remove(p)
        order, cur = [], self.position
        while todo:
            best, nearest = None, []
            for p in todo:
                d = abs(p[1] - cur[1]) + abs(p[0] - cur[0])
                if best is None or d < best:
                    best, nearest = d, []
                if d == best:
                    nearest.append(p)
            cur = self.py_random.choice(nearest)
            order.append(cur)
            todo.remove(cur)
        if not order:
            order.append(self.generate_keypoints(1)[0])
        return order

    # ---- samples --------------------------------------------------------------------------------------
    def detection_cells(self) -> List[Pos]:
        """Cells whose patches feed the detector loss: every box cell plus one random empty cell
        (init_sample, src/env/simple_env.py:397-419)."""
        cells: Set[Pos] = set()
        for box in self.raw_bboxes:
            for p in self.bbox_positions(box):
                cells.add(p)
        empty = [(y, x) for y, x in product(range(self.patch_height), range(self.patch_width)) if (y, x) not in cells]
        if empty:
            cells.add(empty[int(self.rng.choice(len(empty)))])
        return list(cells)

    def generate_sample_indices(self, max_ep_len: int, min_keypoints: int, max_keypoints: int,
                                binomial_keypoints: bool = False, position: Optional[Pos] = None,
                                visited_bbox_patches: Optional[Set[Pos]] = None) -> Dict[str, np.ndarray]:
        """The integer part of ``generate_sample`` (src/env/simple_env.py:481-588): positions [T, 2], current_actions,
        next_actions, labels [T] int64, masks [T] f32, local_bboxes [T, nb, 6] f32, plus positions_yolox [M, 2] and
        bboxes_yolox [M, nb, 6].  Episodes longer than `max_ep_len` keep their LAST `max_ep_len` steps."""
        det_cells = self.detection_cells()
        steps: List[Tuple[int, int, Pos, int, np.ndarray]] = []      # (action taken, teacher action, pos, label, boxes)

        info = self.reset(position, visited_bbox_patches)
        steps.append((Action.LEFT.value, Action.LEFT.value, info["position"], int(info["inside_bbox"]), info["local_bboxes"]))

        def visit(to_visit: Pos, true_target: Pos):
            self.reset(self.position)
            while self.position != to_visit:
                action = move_towards(self.position, to_visit)
                inf = self.step(action)
                best = self.remove_stop_action(move_towards(self.position, true_target))
                self.reset(self.position)
                steps.append((action.value, best.value, inf["position"], int(inf["inside_bbox"]), inf["local_bboxes"]))

        keypoints = self.build_keypoints_trajectory()
        n_extra = int(self.rng.integers(min_keypoints, max_keypoints + 1))
        insert_at = sorted((int(v) for v in self.rng.integers(0, len(keypoints), size=n_extra)), reverse=True)
        for kid, kp in enumerate(keypoints):
            # the teacher action of the last recorded step now points at this keypoint
            a, _, p, lab, lb = steps[-1]
            steps[-1] = (a, self.remove_stop_action(move_towards(self.position, kp)).value, p, lab, lb)
            while kid in insert_at:
                detour = self.generate_binomial_keypoints(1, kp)[0] if binomial_keypoints else self.generate_keypoints(1)[0]
                visit(detour, kp)
                insert_at.remove(kid)
            visit(kp, kp)

        if len(steps) > max_ep_len:
            steps = steps[len(steps) - max_ep_len:]
        T, nb, n = int(max_ep_len), len(self.raw_bboxes), len(steps)
        out = {"positions": np.zeros((T, 2), np.int64), "current_actions": np.zeros(T, np.int64),
               "next_actions": np.zeros(T, np.int64), "labels": np.zeros(T, np.int64), "masks": np.zeros(T, np.float32),
               "local_bboxes": np.zeros((T, nb, 6), np.float32)}
        for i, (a, best, p, lab, lb) in enumerate(steps):
            out["current_actions"][i], out["next_actions"][i], out["labels"][i] = a, best, lab
            out["positions"][i] = p
            out["local_bboxes"][i] = lb
        out["masks"][:n] = 1.0
        out["positions_yolox"] = np.asarray(det_cells, np.int64).reshape(-1, 2)
        out["bboxes_yolox"] = (np.stack([self.local_bboxes(p) for p in det_cells]) if det_cells
                               else np.zeros((0, nb, 6), np.float32)).astype(np.float32)
        return out

    def generate_sample(self, max_ep_len: int, min_keypoints: int, max_keypoints: int, binomial_keypoints: bool = False,
                        position: Optional[Pos] = None, visited_bbox_patches: Optional[Set[Pos]] = None,
                        device=None) -> Dict[str, torch.Tensor]:
        """The reference's sample dict; patches / patches_yolox gathered on the device that holds the image."""
        assert self.image is not None, "generate_sample needs the image (device tensor); use generate_sample_indices"
        idx = self.generate_sample_indices(max_ep_len, min_keypoints, max_keypoints, binomial_keypoints, position,
                                           visited_bbox_patches)
        return assemble_samples(self.image, [self.image_index], [idx], self.patch_size, stacked=False)


def gather_indexed(images: torch.Tensor, image_index: torch.Tensor, positions: torch.Tensor, patch_size: int) -> torch.Tensor:
    """out[n] = images[image_index[n], :, y*P:(y+1)*P, x*P:(x+1)*P]; a negative image index gives a zero patch."""
    from . import _lib
    from ._lib import check, ptr
    assert images.is_cuda and images.dtype == torch.float32 and images.is_contiguous(), "images must be a contiguous f32 device tensor"
    nimg, Cc, H, W = images.shape
    P = int(patch_size)
    assert H % P == 0 and W % P == 0
    ii = image_index.to(images.device, torch.int64).contiguous()
    pos = positions.to(images.device, torch.int64).contiguous()
    N = int(ii.numel())
    if N:
        ph, pw = H // P, W // P
        pc, ic = pos.cpu(), ii.cpu()
        assert bool(((pc[:, 0] >= 0) & (pc[:, 0] < ph) & (pc[:, 1] >= 0) & (pc[:, 1] < pw)).all()), "position outside the grid"
        assert bool((ic < nimg).all()), "image index out of range"
    out = torch.empty((N, Cc, P, P), device=images.device, dtype=torch.float32)
    lib = _lib.load_library()
    if N == 0:                                # empty tensors have no storage to point at
        return out
    check(lib.jn_gather_patches_indexed(ptr(images), ptr(ii), ptr(pos), ptr(out), N, nimg, Cc, H, W, P,
                                        _lib.current_stream(images.device)), "jn_gather_patches_indexed")
    return out


def assemble_samples(images: torch.Tensor, image_ids: Sequence[int], indices: Sequence[Dict[str, np.ndarray]],
                     patch_size: int, stacked: bool = True) -> Dict[str, torch.Tensor]:
    """Turn per-image index dicts into the reference's (collated) tensors with two device gathers: trajectory patches
    [B, T, C, P, P] and detector patches [sum M_i, C, P, P]; boxes are zero-row padded to the longest list
    (NeedleSimpleEnv.collate_fn, src/env/simple_env.py:720-763)."""
    dev = images.device
    B, T = len(indices), indices[0]["masks"].shape[0]
    nb = max(int(d["local_bboxes"].shape[1]) for d in indices)

    def pad(a):
        return a if a.shape[1] == nb else np.concatenate([a, np.zeros((a.shape[0], nb - a.shape[1], 6), np.float32)], 1)

    ii = np.concatenate([np.where(d["masks"] > 0, i, -1) for i, d in zip(image_ids, indices)]).astype(np.int64)
    pos = np.concatenate([d["positions"] for d in indices])
    patches = gather_indexed(images, torch.from_numpy(ii), torch.from_numpy(pos), patch_size)
    iy = np.concatenate([np.full(d["positions_yolox"].shape[0], i, np.int64) for i, d in zip(image_ids, indices)])
    py = np.concatenate([d["positions_yolox"] for d in indices])
    out = {"patches": patches.view(B, T, *patches.shape[1:]),
           "patches_yolox": gather_indexed(images, torch.from_numpy(iy), torch.from_numpy(py), patch_size),
           "bboxes_yolox": torch.from_numpy(np.concatenate([pad(d["bboxes_yolox"]) for d in indices])).to(dev)}
    for k in ("current_actions", "next_actions", "positions", "masks", "labels"):
        out[k] = torch.from_numpy(np.stack([d[k] for d in indices])).to(dev)
    out["local_bboxes"] = torch.from_numpy(np.stack([pad(d["local_bboxes"]) for d in indices])).to(dev)
    if not stacked:
        assert B == 1
        out = {k: (v if k in ("patches_yolox", "bboxes_yolox") else v[0]) for k, v in out.items()}
    return out
