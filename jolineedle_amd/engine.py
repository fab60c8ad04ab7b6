"""Owner of one ``jn_ctx`` (include/jnroll.h): creation from the reference's config
objects, state-dict table, weight upload.  Shared by GPT / NeedleYOLOX / the env."""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import JnConfig, JnParamInfo, JnTensor, check

# src/models/gpt.py:190-218
GPT_ZOO = {
    "openai-gpt": (12, 12, 768), "gpt2": (12, 12, 768), "gpt2-medium": (24, 16, 1024),
    "gpt2-large": (36, 20, 1280), "gpt2-xl": (48, 25, 1600), "gopher-44m": (8, 16, 512),
    "gpt-mini": (6, 6, 192), "gpt-micro": (4, 4, 128), "gpt-nano": (3, 3, 48), "gpt-pico": (2, 2, 32),
}
# (depth, width, depthwise) behind the yolox_* factories (src/models/gpt.py:242-250)
YOLOX_SIZES = {
    "yolox": (0.33, 0.25, True), "yolox-nano": (0.33, 0.25, True), "yolox-tiny": (0.33, 0.375, False),
    "yolox-s": (0.33, 0.50, False), "yolox-m": (0.67, 0.75, False), "yolox-l": (1.0, 1.0, False),
    "yolox-x": (1.33, 1.25, False),
}


class Engine:
    def __init__(self, jn_cfg: JnConfig):
        self.lib = _lib.load_library()
        self.cfg = jn_cfg
        h = C.c_void_p()
        check(self.lib.jn_create(C.byref(jn_cfg), C.byref(h)), "jn_create")
        self.handle = h
        self._keep = None

    def close(self):
        if getattr(self, "handle", None):
            self.lib.jn_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def param_table(self):
        n = self.lib.jn_param_count(self.handle)
        out = []
        for i in range(n):
            info = JnParamInfo()
            check(self.lib.jn_param_info_at(self.handle, i, C.byref(info)), "jn_param_info_at")
            shape = tuple(info.shape[k] for k in range(info.ndim))
            out.append((info.name.decode(), shape, int(info.dtype), bool(info.is_buffer), bool(info.used)))
        return out

    def load_state_dict(self, sd):
        """Upload a reference-named state dict (host copies are made as fp32 / int64)."""
        keep, arr = [], (JnTensor * len(sd))()
        for i, (name, t) in enumerate(sd.items()):
            t = t.detach().to("cpu")
            t = (t.to(torch.int64) if t.dtype in (torch.int64, torch.int32) else t.to(torch.float32)).contiguous()
            keep.append((name.encode(), t))
            arr[i].name = keep[-1][0]
            arr[i].data = t.data_ptr()
            arr[i].dtype = 1 if t.dtype == torch.int64 else 0
            arr[i].ndim = t.dim()
            for k, s in enumerate(t.shape[:4]):
                arr[i].shape[k] = s
        check(self.lib.jn_load_weights(self.handle, arr, len(sd)), "jn_load_weights")


def make_jn_config(config, device_index, max_batch, n_actions):
    """reference model_config (main.py:367-386) -> jn_config."""
    if getattr(config, "model_type", None) is not None:
        n_layer, n_head, n_embd = GPT_ZOO[config.model_type]
    else:
        n_layer, n_head, n_embd = config.n_layer, config.n_head, config.n_embd
    c = JnConfig()
    c.struct_size = C.sizeof(JnConfig)
    c.device = device_index
    c.n_layer, c.n_head, c.n_embd = n_layer, n_head, n_embd
    c.block_size = config.block_size
    c.n_actions = n_actions
    c.patch_size = config.patch_size
    c.use_pos_emb = int(bool(config.use_pos_emb))
    c.no_patch_emb = int(bool(config.no_patch_emb))
    c.concat_emb = int(bool(config.concat_emb))
    c.decoder_pos_encoding = int(bool(config.decoder_pos_encoding))
    c.pos_emb_size = int(getattr(config, "pos_emb_size", 1) or 1)
    gb = getattr(config, "gpt_backbone", None)
    if gb:
        c.gpt_bb_depth, c.gpt_bb_width, dw = YOLOX_SIZES[gb]
        c.gpt_bb_depthwise = int(dw)
    ip = getattr(config, "image_processor", None)
    if ip and getattr(config, "with_detector", True):
        c.with_detector = 1
        c.det_depth, c.det_width, dw = YOLOX_SIZES[ip]
        c.det_depthwise = int(dw)
    c.det_conf_threshold = float(getattr(config, "detector_conf_threshold", 0.5))
    c.det_nms_threshold = 0.45
    c.max_batch = max_batch
    c.max_det_per_patch = int(getattr(config, "max_det_per_patch", 64))
    c.act_dtype = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}[str(getattr(config, "act_dtype", "f32"))]
    return c


def bare_env_config(patch_size, max_batch, device_index, block_size):
    """Context that only serves the environment entry points (no networks)."""
    c = JnConfig()
    c.struct_size = C.sizeof(JnConfig)
    c.device = device_index
    c.n_layer, c.n_head, c.n_embd = 1, 1, 4
    c.block_size, c.n_actions, c.patch_size = max(1, min(block_size, 255)), 9, patch_size
    c.no_patch_emb = 1
    c.max_batch = max_batch
    return c
