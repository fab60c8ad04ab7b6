"""ctypes binding of libjnroll.so (the drop-in boundary, include/jnroll.h).

Fails loudly when the library is missing or a call returns an error: there is no
Python/CPU fallback for any entry point.
"""
import ctypes as C
import os
from pathlib import Path

_LIB = None
LIB_PATH = Path(__file__).resolve().parent / "lib" / "libjnroll.so"

JN_MODE_GREEDY, JN_MODE_SAMPLE, JN_MODE_FORCED = 0, 1, 2
JN_NET_GPT_BACKBONE, JN_NET_DETECTOR = 0, 1


class LibraryNotBuilt(RuntimeError):
    pass


class JnError(RuntimeError):
    pass


class JnConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32),
                ("n_layer", C.c_int32), ("n_head", C.c_int32), ("n_embd", C.c_int32),
                ("block_size", C.c_int32), ("n_actions", C.c_int32), ("patch_size", C.c_int32),
                ("use_pos_emb", C.c_int32), ("no_patch_emb", C.c_int32), ("concat_emb", C.c_int32),
                ("decoder_pos_encoding", C.c_int32), ("pos_emb_size", C.c_int32),
                ("gpt_bb_depth", C.c_float), ("gpt_bb_width", C.c_float), ("gpt_bb_depthwise", C.c_int32),
                ("with_detector", C.c_int32), ("det_depth", C.c_float), ("det_width", C.c_float),
                ("det_depthwise", C.c_int32), ("det_conf_threshold", C.c_float),
                ("det_nms_threshold", C.c_float), ("max_batch", C.c_int32), ("max_det_per_patch", C.c_int32),
                ("act_dtype", C.c_int32)]


class JnTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4)]


class JnParamInfo(C.Structure):
    _fields_ = [("name", C.c_char * 160), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4), ("is_buffer", C.c_int32), ("used", C.c_int32)]


class JnTrainOpts(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("reward_norm", C.c_int32), ("ret_mean", C.c_float),
                ("ret_std", C.c_float), ("entropy_weight", C.c_float), ("loss_scale", C.c_float)]


class JnRolloutOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "rewards_dev", "returns_dev", "logprobs_dev", "entropies_dev", "masks_dev", "logit_masks_dev",
        "positions_dev", "actions_dev", "logits_dev", "final_emb_dev", "patches_dev", "det_boxes_dev",
        "det_counts_dev")]


ABI_VERSION = 2      # JN_ABI_VERSION of include/jnroll.h

# name -> (restype, argtypes); every symbol include/jnroll.h declares
SIGNATURES = {
    "jn_abi_version": (C.c_int, []),
    "jn_last_error": (C.c_char_p, []),
    "jn_create": (C.c_int, [C.POINTER(JnConfig), C.POINTER(C.c_void_p)]),
    "jn_destroy": (C.c_int, [C.c_void_p]),
    "jn_param_count": (C.c_int, [C.c_void_p]),
    "jn_param_info_at": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(JnParamInfo)]),
    "jn_load_weights": (C.c_int, [C.c_void_p, C.POINTER(JnTensor), C.c_size_t]),
    "jn_env_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_void_p]),
    "jn_env_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "jn_env_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jn_env_state": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "jn_env_patches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "jn_gather_patches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p]),
    "jn_augment_patches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p]),
    "jn_gather_patches_indexed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "jn_backbone_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "jn_read_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "jn_embed_patches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "jn_reinforce_step": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                                    C.POINTER(JnTrainOpts), C.POINTER(JnRolloutOut), C.c_void_p, C.c_void_p]),
    "jn_supervised_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jn_supervised_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "jn_supervised_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "jn_optimizer_step": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "jn_arena_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "jn_set_grad_arena": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "jn_read_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "jn_zero_grad": (C.c_int, [C.c_void_p, C.c_void_p]),
    "jn_backbone_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "jn_read_grad": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "jn_gpt_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jn_detect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jn_rollout": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int,
                             C.POINTER(JnRolloutOut), C.c_void_p]),
    "jn_rollout_steps": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_void_p]),
    "jn_last_timing": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "jn_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "jn_detector_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "jn_detector_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jn_detector_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p]),
    "jn_optimizer_step_group": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "jn_reinforce_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                                       C.POINTER(JnRolloutOut), C.c_void_p]),
    "jn_reinforce_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jn_arena_segment": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "jn_export_arena": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "jn_import_arena": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "jn_set_dropout": (C.c_int, [C.c_void_p, C.c_float, C.c_uint64]),
    "jn_set_freeze": (C.c_int, [C.c_void_p, C.c_int]),
    "jn_optimizer_steps": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]),
}


def load_library(path=None):
    """dlopen libjnroll.so and type every entry point.  Raises LibraryNotBuilt if absent."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = Path(path or os.environ.get("JNROLL_LIB", LIB_PATH))
    if not p.exists():
        raise LibraryNotBuilt(
            f"{p} not found: the HIP extension is not built. Run `bash jolineedle_amd/csrc/build.sh` "
            "(or __graft_entry__.build()). There is no CPU fallback for this path.")
    lib = C.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.jn_abi_version() != ABI_VERSION:
        raise JnError(f"libjnroll ABI {lib.jn_abi_version()} != {ABI_VERSION}")
    if path is None:
        _LIB = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load_library().jn_last_error().decode(errors="replace")
        raise JnError(f"{what} failed with code {rc}: {msg}")


def ptr(t):
    """Device/host pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
