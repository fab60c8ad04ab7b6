"""Configuration surface of the reference's entry point for this path.

``CfgNode`` mirrors src/utils.py:15-92 (attribute bag); ``get_args`` keeps the flag
names of main.py:24-307 that shape the rollout / model (the dataset, Visdom and
checkpoint flags are accepted and carried so reference command lines parse);
``args_to_config`` follows main.py:310-388.
"""
import argparse
import math
from ast import literal_eval


class CfgNode:
    """Nested attribute bag with the interface of the reference's ``CfgNode`` (src/utils.py:15-92): keyword construction,
    ``str()`` as an indented listing, ``to_dict`` / ``merge_from_dict``, and ``merge_from_args(["--a.b=value", ...])``
    overrides of attributes that already exist (values parsed as Python literals when they are ones)."""

    def __init__(self, **entries):
        vars(self).update(entries)

    def _lines(self, depth=0):
        pad = " " * (4 * depth)
        for name, value in vars(self).items():
            if isinstance(value, CfgNode):
                yield f"{pad}{name}:\n"
                yield from value._lines(depth + 1)
            else:
                yield f"{pad}{name}: {value}\n"

    def __str__(self):
        return "".join(self._lines())

    def to_dict(self):
        return {name: value.to_dict() if isinstance(value, CfgNode) else value for name, value in vars(self).items()}

    def merge_from_dict(self, d):
        vars(self).update(d)

    def merge_from_args(self, args):
        for override in args:
            flag, sep, text = override.partition("=")
            if not sep or "=" in text or not flag.startswith("--"):
                raise AssertionError(f"override must look like --name=value or --node.name=value, got {override!r}")
            try:
                value = literal_eval(text)
            except (ValueError, SyntaxError):
                value = text                                # a plain string
            *path, leaf = flag[2:].split(".")
            node = self
            for part in path:
                node = getattr(node, part)
            if not hasattr(node, leaf):
                raise AssertionError(f"{flag[2:]} is not an attribute that exists in the config")
            setattr(node, leaf, value)


def get_args(args=None):
    p = argparse.ArgumentParser(description="MinGPT Needle (MI355X rollout engine)")
    p.add_argument("--training-mode", type=str, default="supervised", choices=["supervised", "reinforce"])
    p.add_argument("--model-type", type=str, default="gpt-mini")
    p.add_argument("--max-seq-len", type=int, default=32)
    p.add_argument("--test-max-seq-len", type=int)
    p.add_argument("--patch-size", type=int, default=224)
    p.add_argument("--minimum-image-size", type=int, default=224 * 5)
    p.add_argument("--no-detection", action="store_false", dest="detection_enabled")
    p.add_argument("--image-processor", type=str, default="yolox")
    p.add_argument("--gpt-backbone", type=str, default=None)
    p.add_argument("--freeze-image-processor", action="store_true")
    p.add_argument("--detector-conf-threshold", type=float, default=0.5)
    p.add_argument("--use-positional-embedding", action="store_true")
    p.add_argument("--no-patch-embedding", action="store_true")
    p.add_argument("--concat-embeddings", action="store_true")
    p.add_argument("--decoder-pos-encoding", action="store_true")
    p.add_argument("--dropout", type=float, default=0.1)
    p.add_argument("--enable-stop", action="store_true")
    p.add_argument("--weight-decay", type=float, default=0.0)
    p.add_argument("--stop-weight", type=float, default=1.0)
    p.add_argument("--no-reward-norm", action="store_false", dest="reward_norm")
    p.add_argument("--entropy-weight", type=float, default=0.01)
    p.add_argument("--binomial-keypoints", action="store_true")
    p.add_argument("--min-keypoints", type=int, default=0)
    p.add_argument("--max-keypoints", type=int, default=0)
    p.add_argument("--merge-bboxes", action="store_true")
    p.add_argument("--loss", type=str, default="on-optimal-trajectory")
    p.add_argument("--yolo-lr", type=float, default=1e-4)
    p.add_argument("--augment-rotate", action="store_true")
    p.add_argument("--augment-translate", action="store_true")
    p.add_argument("--devices", nargs="+", type=int)
    p.add_argument("--port-ddp", type=int, default=12355)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--max-iters", type=int, default=1000)
    p.add_argument("--batch-size", type=int, default=8)
    p.add_argument("--gradient-accumulation", type=int, default=1)
    p.add_argument("--env-name", type=str, default="test")
    p.add_argument("--group", type=str, default="")
    p.add_argument("--work-dir", type=str, default="./out/")
    p.add_argument("--test-size", type=float, default=0.01)
    p.add_argument("--test-samples", type=int, default=100)
    p.add_argument("--test-pattern", type=str, default="")
    p.add_argument("--test-every", type=int, default=500)
    p.add_argument("--failure-select-rate", type=float, default=0.1)
    p.add_argument("--eval-training-set", action="store_true")
    p.add_argument("--resume-training", type=str, default=None)
    p.add_argument("--detection-checkpoint", type=str, default=None)
    p.add_argument("--dataset-dir", type=str, default=None)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--train-size", type=int, default=-1)
    p.add_argument("--num-workers", type=int, default=1)
    p.add_argument("--generated-sample-eval-size", type=int, default=500)
    p.add_argument("--filter-classes", nargs="+", default=None)
    p.add_argument("--measure-flops", action="store_true")
    p.add_argument("--no-recurrent-embedding", action="store_true")
    return p.parse_args(args)


def args_to_config(args):
    """-> (train_config, model_config), field names of main.py:310-388."""
    t = CfgNode()
    t.training_mode = args.training_mode
    t.learning_rate, t.yolo_lr, t.weight_decay = args.lr, args.yolo_lr, args.weight_decay
    t.max_iters, t.batch_size = args.max_iters, args.batch_size
    t.detection_enabled = args.detection_enabled
    t.gradient_accumulation = args.gradient_accumulation
    t.env_name, t.work_dir, t.seed = args.env_name, args.work_dir, args.seed
    t.test_every, t.test_samples = args.test_every, args.test_samples
    t.stop_weight, t.entropy_weight, t.reward_norm = args.stop_weight, args.entropy_weight, args.reward_norm
    t.merge_bboxes, t.failure_select_rate = args.merge_bboxes, args.failure_select_rate
    t.port_ddp = args.port_ddp
    t.gpu_ids = args.devices if args.devices else [0]
    t.world_size = len(t.gpu_ids)
    t.max_seq_len = args.max_seq_len
    t.test_max_seq_len = args.test_max_seq_len if args.test_max_seq_len else args.max_seq_len
    t.patch_size, t.n_channels = args.patch_size, 3
    t.stop_enabled = args.enable_stop
    t.image_cols = math.ceil(2064 / t.patch_size)       # main.py:364-366
    m = CfgNode()
    m.model_type, m.n_layer, m.n_head, m.n_embd = args.model_type, None, None, None
    m.embd_pdrop = m.resid_pdrop = m.attn_pdrop = 0.1
    m.image_processor, m.gpt_backbone = args.image_processor, args.gpt_backbone
    m.freeze_image_processor = args.freeze_image_processor
    m.detector_conf_threshold = args.detector_conf_threshold
    m.use_pos_emb, m.no_patch_emb = args.use_positional_embedding, args.no_patch_embedding
    m.concat_emb, m.decoder_pos_encoding = args.concat_embeddings, args.decoder_pos_encoding
    m.pos_emb_size = t.image_cols ** 2
    m.dropout = args.dropout
    m.block_size, m.n_channels, m.patch_size, m.image_cols = t.max_seq_len, 3, t.patch_size, t.image_cols
    m.no_recurrent_embedding = args.no_recurrent_embedding
    return t, m


def model_config(**kw):
    """Reference-style ``model_config`` node (main.py:367-386) with the defaults of BASELINE configs[2] (gpt-nano +
    yolox-nano patch encoder + yolox-s detector, 448 px, block size 20); keyword arguments override fields,
    ``nclasses`` sizes the categorical action head."""
    from .common import ActionInfo
    cfg = dict(model_type="gpt-nano", n_layer=None, n_head=None, n_embd=None, block_size=20, patch_size=448,
               image_processor="yolox-s", gpt_backbone="yolox-nano", use_pos_emb=True, no_patch_emb=False,
               concat_emb=True, decoder_pos_encoding=True, pos_emb_size=25, dropout=0.0,
               detector_conf_threshold=0.5, no_recurrent_embedding=False, with_detector=True, nclasses=9)
    cfg.update(kw)
    n = cfg.pop("nclasses")
    cfg["actions_info"] = [ActionInfo("categorical", n)]
    return CfgNode(**cfg)
