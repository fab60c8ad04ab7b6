"""``NeedleGeneralEnv`` — batched needle environment of the reference
(src/env/general_env.py:14-573) with all state on the device, stepped by
libjnroll.so (``jn_env_*``).  n_glimps_levels must be 1 (src/reinforce.py:58)."""
from typing import Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import check, ptr
from .engine import Engine, bare_env_config


class NeedleGeneralEnv:
    def __init__(self, images: Tensor, bboxes: Tensor, patch_size: int, max_ep_len: int,
                 n_glimps_levels: int = 1, stop_enabled: bool = False, engine: Engine = None):
        assert images.shape[0] == bboxes.shape[0]          # general_env.py:37-39
        assert len(images.shape) == 4
        assert n_glimps_levels > 0
        if n_glimps_levels != 1:
            raise NotImplementedError("only n_glimps_levels == 1 is on the rollout path (src/reinforce.py:58)")
        if not images.is_cuda:
            raise RuntimeError("NeedleGeneralEnv of the HIP engine needs images on the GPU (no CPU fallback)")
        self.patch_size, self.max_ep_len = patch_size, max_ep_len
        self.n_glimps_levels, self.stop_enabled = n_glimps_levels, stop_enabled
        self.batch_size, self.n_channels, self.height, self.width = images.shape
        assert self.n_channels == 3
        assert self.height % self.patch_size == 0          # general_env.py:50-51
        assert self.width % self.patch_size == 0
        self.n_vertical_patches = self.height // patch_size
        self.n_horizontal_patches = self.width // patch_size
        self.device = images.device
        self._images = images.to(torch.float32).contiguous()
        self.bboxes = bboxes
        self._bboxes_dev = bboxes.to(self.device, torch.int64).contiguous()
        self._engine = None
        self.bind(engine)

    # ---- engine binding ----------------------------------------------------------------
    def bind(self, engine: Engine = None):
        """(Re-)create the device state inside `engine` (the model's context for rollouts)."""
        if engine is None:
            engine = Engine(bare_env_config(self.patch_size, self.batch_size, self.device.index or 0,
                                            self.max_ep_len))
        if engine is self._engine:
            return
        self._engine = engine
        nb = self._bboxes_dev.shape[1] if self._bboxes_dev.dim() == 3 else 0
        check(engine.lib.jn_env_init(engine.handle, ptr(self._images), ptr(self._bboxes_dev), self.batch_size,
                                     self.height, self.width, nb, self.max_ep_len, int(self.stop_enabled),
                                     self._stream()), "jn_env_init")

    def _stream(self):
        return _lib.current_stream(self.device)

    def _view(self, what, shape, dtype):
        import ctypes as C
        p = C.c_void_p()
        check(self._engine.lib.jn_env_state(self._engine.handle, what, C.byref(p)), "jn_env_state")
        n = 1
        for s in shape:
            n *= s
        itemsize = torch.empty((), dtype=dtype).element_size()
        out = torch.empty(shape, dtype=dtype, device=self.device)
        from .hipmem import copy_d2d
        copy_d2d(out.data_ptr(), p.value, n * itemsize, self.device)
        return out

    # ---- reference surface -------------------------------------------------------------
    @property
    def images(self) -> Tensor:
        return self._images.unsqueeze(1)                   # [B, 1, C, H, W] (general_env.py:115)

    @property
    def positions(self) -> Tensor:
        return self._view(0, (self.batch_size, 2), torch.int64)

    @property
    def bbox_masks(self) -> Tensor:
        g = (self.batch_size, self.n_vertical_patches, self.n_horizontal_patches)
        return self._view(1, g, torch.uint8).bool()

    @property
    def visited_patches(self) -> Tensor:
        g = (self.batch_size, self.n_vertical_patches, self.n_horizontal_patches)
        return self._view(2, g, torch.uint8).bool()

    @property
    def steps(self) -> Tensor:
        return self._view(3, (self.batch_size,), torch.int32).long()

    @property
    def has_stopped(self) -> Tensor:
        return self._view(4, (self.batch_size,), torch.uint8).bool()

    @property
    def patches(self) -> Tensor:
        out = torch.empty((self.batch_size, 3, self.patch_size, self.patch_size), device=self.device)
        check(self._engine.lib.jn_env_patches(self._engine.handle, ptr(out), self._stream()), "jn_env_patches")
        return out.unsqueeze(1)                            # [B, glimps_level = 1, C, P, P]

    def reset(self, positions=None, seed: int = 0) -> Tuple[Tensor, dict]:
        if positions is not None:
            positions = positions.to(self.device, torch.int64).contiguous()
        check(self._engine.lib.jn_env_reset(self._engine.handle, ptr(positions), seed, self._stream()), "jn_env_reset")
        return self.patches, {"positions": self.positions}

    @torch.no_grad()
    def step(self, actions: Tensor):
        actions = actions.to(self.device, torch.int64).contiguous()
        B = self.batch_size
        rewards = torch.empty((B,), device=self.device, dtype=torch.float32)
        term = torch.empty((B,), device=self.device, dtype=torch.uint8)
        trunc = torch.empty((B,), device=self.device, dtype=torch.uint8)
        check(self._engine.lib.jn_env_step(self._engine.handle, ptr(actions), ptr(rewards), ptr(term), ptr(trunc),
                                           self._stream()), "jn_env_step")
        return self.patches, rewards, term.bool(), trunc.bool(), {"positions": self.positions}

    @property
    def terminated(self) -> Tensor:
        if self.stop_enabled:
            return self.has_stopped
        m, v = self.bbox_masks, self.visited_patches
        return ((m & v) != m).sum(dim=(1, 2)) == 0

    @property
    def prop_patches_found(self) -> Tensor:
        m, v = self.bbox_masks, self.visited_patches
        count = (m & v).sum(dim=(1, 2))
        tot = m.sum(dim=(1, 2))
        tot[tot == 0] = 1
        return count / tot

    # ---- detection bookkeeping (src/env/general_env.py:381-573) ----------------------------------------
    def parse_bboxes(self, bboxes: Tensor = None):
        """Ground-truth boxes split over the patch grid: ([B, Gy, Gx, nb, 4] patch-local xyxy, [B, Gy, Gx, nb] masks)."""
        from .detection import split_bboxes_over_patches
        return split_bboxes_over_patches(self.bboxes if bboxes is None else bboxes, self.n_vertical_patches,
                                         self.n_horizontal_patches, self.patch_size)

    def get_detection_targets(self):
        from .detection import detection_targets
        return detection_targets(self.bboxes, self.n_vertical_patches, self.n_horizontal_patches, self.patch_size)

    @torch.no_grad()
    def get_detection_batch(self, sample_neg: int = 1, generator: torch.Generator = None):
        """Patches to train the detector on: every patch holding (a piece of) a box plus `sample_neg` random empty
        patches per image; returns (patches [n, 3, P, P], bboxes [n, nb, 1 + 4]) like general_env.py:503-544."""
        boxes, masks = self.parse_bboxes()
        any_box = masks.any(-1).cpu()
        P = self.patch_size
        patches, all_boxes = [], []
        for i in range(self.batch_size):
            pos = torch.nonzero(any_box[i])
            neg = torch.nonzero(~any_box[i])
            neg = neg[torch.randperm(len(neg), generator=generator)[:sample_neg]]
            for y, x in torch.cat((pos, neg)).tolist():
                patches.append(self._images[i, :, y * P:(y + 1) * P, x * P:(x + 1) * P])
                all_boxes.append(torch.nn.functional.pad(boxes[i, y, x], (1, 0)))
        return torch.stack(patches), torch.stack(all_boxes).to(self.device)

    @property
    def prop_bboxes_found(self) -> Tensor:
        return (self.prop_patches_found > 0).to(torch.float32)
