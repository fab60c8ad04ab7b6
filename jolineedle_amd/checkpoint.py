"""Checkpoint / ``config.json`` compatibility: ``main.save_config`` (main.py:436-449),
``main.load_checkpoint`` (:532-563), ``main.load_detection_checkpoint`` (:566-584) and ``infer.config_from_file``
(infer.py:58-73) for the accelerated model — same file names (``config.json``, ``checkpoint.pt``, ``checkpoint_best.pt``),
same dictionary keys (``model`` with optional DDP ``module.`` prefixes), so a directory written by the reference loads.

Deviation: AdamW moments live inside the engine and are not part of the C ABI; ``optimizer-gpt`` / ``optimizer-yolox``
entries of a reference checkpoint are ignored (the moments restart) and checkpoints written here carry empty ones."""
import json
from copy import deepcopy
from pathlib import Path
from typing import Tuple

import torch

from .config import CfgNode


def _plain(v):
    if isinstance(v, CfgNode):
        return {k: _plain(x) for k, x in vars(v).items()}
    if isinstance(v, (set, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, list):
        return [_plain(x) for x in v]
    if hasattr(v, "__dataclass_fields__"):
        return {k: _plain(getattr(v, k)) for k in v.__dataclass_fields__}
    return v


def save_config(model_config, train_config, folder=None) -> Path:
    """``<work_dir>/<env_name>/config.json`` with the two sections of main.py:436-449."""
    folder = Path(folder) if folder is not None else Path(train_config.work_dir) / train_config.env_name
    folder.mkdir(parents=True, exist_ok=True)
    path = folder / "config.json"
    with open(path, "w") as f:
        json.dump({"model": _plain(deepcopy(model_config)), "train": _plain(deepcopy(train_config))}, f, indent=4)
    return path


def config_from_file(config_path) -> Tuple[CfgNode, CfgNode]:
    """(train_config, model_config) from a ``config.json`` (infer.py:58-73): defaults overridden key by key."""
    from .gpt import GPT
    with open(config_path) as f:
        cj = json.load(f)
    train_config = CfgNode()
    for k, v in cj.get("train", {}).items():
        setattr(train_config, k, v)
    model_config = GPT.get_default_config()
    for k, v in cj.get("model", {}).items():
        if k == "actions_info":
            continue                       # rebuilt from the train config (infer.py:85-86), never read from the file
        setattr(model_config, k, v)
    return train_config, model_config


def _strip_ddp(sd):
    return {k.replace("module.", ""): v for k, v in sd.items()}


def save_checkpoint(model, folder, best: bool = False, extra: dict = None) -> Path:
    folder = Path(folder)
    folder.mkdir(parents=True, exist_ok=True)
    if hasattr(model, "pull_parameters") and getattr(model, "_engine", None) is not None:
        model.pull_parameters()            # optimiser-updated weights and BN statistics live in the engine
    ck = {"model": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "optimizer-gpt": {}, "optimizer-yolox": {}}
    ck.update(extra or {})
    path = folder / ("checkpoint_best.pt" if best else "checkpoint.pt")
    torch.save(ck, path)
    return path


def load_checkpoint(train_config, trainer_or_model, best: bool = False) -> None:
    """main.py:532-563: ``train_config.resume_training`` names the directory."""
    model = getattr(trainer_or_model, "model", trainer_or_model)
    folder = Path(train_config.resume_training)
    ck = torch.load(folder / ("checkpoint_best.pt" if best else "checkpoint.pt"), map_location="cpu", weights_only=False)
    model.load_state_dict(_strip_ddp(ck["model"]))


def load_detection_checkpoint(train_config, trainer_or_model) -> None:
    """main.py:566-584: only the ``yolox.*`` entries of another run's checkpoint replace the detector."""
    model = getattr(trainer_or_model, "model", trainer_or_model)
    ck = torch.load(train_config.detection_checkpoint, map_location="cpu", weights_only=False)
    det = {k: v for k, v in _strip_ddp(ck["model"]).items() if k.startswith("yolox.")}
    assert det, "the detection checkpoint holds no yolox.* tensors"
    sd = model.state_dict()
    missing = [k for k in sd if k.startswith("yolox.") and k not in det]
    assert not missing, f"detection checkpoint lacks {missing[:3]} ..."
    sd.update(det)
    model.load_state_dict(sd)
