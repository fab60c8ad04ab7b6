"""jolineedle_amd — MI355X-native glimpse-rollout engine behind JoliNeedle's operator API.

Python mirror of the reference interface for the rollout hot path:
``GPT``, ``NeedleYOLOX``, ``NeedleGeneralEnv``, ``ReinforceTrainer.rollout`` with the
reference's names, argument meaning and error behaviour, over the C ABI of
``libjnroll.so`` (include/jnroll.h).  There is no CPU fallback: the HIP library must be
built (``python -c 'import __graft_entry__ as g; g.build()'``) and a GPU present.
"""
from .common import Action, ACTION_DELTAS, ActionInfo, get_actions_info  # noqa: F401
from .config import CfgNode, get_args, args_to_config  # noqa: F401
from ._lib import load_library, LibraryNotBuilt  # noqa: F401
from .gpt import GPT  # noqa: F401
from .env import NeedleGeneralEnv  # noqa: F401
from .yolox import NeedleYOLOX  # noqa: F401
from .reinforce import ReinforceTrainer  # noqa: F401
from .supervised import SupervisedTrainer  # noqa: F401
from .data import padded_collate, synthetic_batch  # noqa: F401
from .augment import DetectionAugment  # noqa: F401
from .checkpoint import (save_config, config_from_file, save_checkpoint, load_checkpoint,  # noqa: F401
                         load_detection_checkpoint)
from .infer import infer_images, pad_to_patch_multiple, load_bboxes  # noqa: F401
from .trajectory import NeedleSimpleEnv  # noqa: F401
from .detection import (patch_bboxes2full_image, merge_boxes, merge_boxes_batched,  # noqa: F401
                        compute_detection_metrics, detection_targets)

__all__ = ["GPT", "NeedleYOLOX", "NeedleGeneralEnv", "ReinforceTrainer", "SupervisedTrainer", "Action", "ACTION_DELTAS",
           "ActionInfo", "get_actions_info", "CfgNode", "get_args", "args_to_config", "load_library"]
