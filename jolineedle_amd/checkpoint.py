"""Checkpoint / ``config.json`` compatibility: ``main.save_config`` (main.py:436-449),
``main.load_checkpoint`` (:532-563), ``main.load_detection_checkpoint`` (:566-584) and ``infer.config_from_file``
(infer.py:58-73) for the accelerated model — same file names (``config.json``, ``checkpoint.pt``, ``checkpoint_best.pt``),
same dictionary keys (``model`` with optional DDP ``module.`` prefixes), so a directory written by the reference loads.

``optimizer-gpt`` / ``optimizer-yolox`` hold ``torch.optim.AdamW`` state dicts in both directions: the engine's moments
and step counters are exported in the reference's tensor layout (``EngineAdamW.state_dict``), and a reference
checkpoint's moments are imported on resume, so a resumed run continues with the bias correction it stopped at."""
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


def save_checkpoint(model, folder, best: bool = False, extra: dict = None, optim_gpt=None, optim_yolox=None,
                    train_config=None) -> Path:
    """main.py:436-449's checkpoint dict.  `optim_gpt` / `optim_yolox`: the optimisers of ``model.configure_optimizers``.
    Without them the model's most recent optimisers are used, else fresh front ends onto the engine's state (where the
    moments live) carrying `train_config`'s learning rates — the file's ``param_groups`` are what a resumed run trains
    with (``AdamW.load_state_dict`` restores lr), so they must hold real values."""
    folder = Path(folder)
    folder.mkdir(parents=True, exist_ok=True)
    if hasattr(model, "pull_parameters") and getattr(model, "_engine", None) is not None:
        model.pull_parameters()            # optimiser-updated weights and BN statistics live in the engine
    if optim_gpt is None and getattr(model, "_last_optimizers", None) is not None:
        optim_gpt, optim_yolox = model._last_optimizers
    if optim_gpt is None and hasattr(model, "configure_optimizers") and torch.cuda.is_available():
        from .config import CfgNode as _C
        # learning rates for the file's param_groups: the train config, else what the engine's optimiser last stepped
        # with (train_iteration records it), else torch.optim.AdamW's default — never 0, which a resumed run would train with
        last = getattr(model, "_last_lr", None) or (1e-3, 1e-3)
        lr = float(getattr(train_config, "learning_rate", 0.0) or 0.0) if train_config is not None else 0.0
        lr = lr if lr > 0.0 else float(last[0])
        ylr = float(getattr(train_config, "yolo_lr", 0.0) or 0.0) if train_config is not None else 0.0
        ylr = ylr if ylr > 0.0 else float(last[1])
        keep = getattr(model, "_last_optimizers", None)
        optim_gpt, optim_yolox = model.configure_optimizers(_C(learning_rate=lr, yolo_lr=ylr))
        object.__setattr__(model, "_last_optimizers", keep)
    ck = {"model": {k: v.detach().cpu() for k, v in model.state_dict().items()},
          "optimizer-gpt": optim_gpt.state_dict() if optim_gpt is not None else {},
          "optimizer-yolox": optim_yolox.state_dict() if optim_yolox is not None else {}}
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
    # optimiser states (main.py:548-556): the trainer's own optimisers when it has them, else front ends onto the engine
    og, oy = getattr(trainer_or_model, "optim_gpt", None), getattr(trainer_or_model, "optim_yolox", None)
    if og is None and torch.cuda.is_available() and (ck.get("optimizer-gpt") or ck.get("optimizer-yolox")):
        from .config import CfgNode as _C
        og, oy = model.configure_optimizers(_C(learning_rate=float(getattr(train_config, "learning_rate", 0.0) or 0.0),
                                               yolo_lr=float(getattr(train_config, "yolo_lr", 0.0) or 0.0)))
    model.sync_weights()
    if og is not None and ck.get("optimizer-gpt"):
        og.load_state_dict(ck["optimizer-gpt"])
    if oy is not None and ck.get("optimizer-yolox"):
        oy.load_state_dict(ck["optimizer-yolox"])


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
