"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own
Python (imported read-only from /root/reference) on CPU.

Run in the build container only:  python tests/golden/make_golden.py
(/root/reference does not exist on the GPU box; tests only read the .npz files.)

The reference's third-party imports that are absent here (yolox,
positional_encodings, kornia, torchvision, gymnasium, torchmetrics, cv2, thop,
torchinfo, visdom) are replaced by name-only stub modules in ``sys.modules``.
Where the hot path calls into them (YOLOX nets, sinusoid tables, Boxes.to_mask,
nms) the stub forwards to the oracle's restatement, so those parts remain
"parity unpinned" (oracle/__init__.py); everything else that executes below is
the reference's own code: GPT (+Block/attention/GELU/embeddings/recurrence),
ActionHead, NeedleGeneralEnv, ReinforceTrainer.rollout / sample_from_logits /
compute_metrics, Trainer.patch_bboxes2full_image.

Fixtures hold inputs, seeds and expected outputs only (no reference source).
"""
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, str(REPO))

from oracle import yolox_ref, posenc_ref  # noqa: E402
from oracle.gpt_ref import build_gpt_ref  # noqa: E402
from tests.helpers import randomize_bn, synth_batch, synth_tokens  # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    # --- yolox -------------------------------------------------------------------
    def factory(name):
        def make(pretrained=False, num_classes=80, device="cpu"):
            return yolox_ref.build_yolox(name, num_classes)
        return make

    models = _mod("yolox.models", YOLOX=yolox_ref.YOLOX,
                  yolox_nano=factory("yolox-nano"), yolox_tiny=factory("yolox-tiny"),
                  yolox_s=factory("yolox-s"), yolox_m=factory("yolox-m"),
                  yolox_l=factory("yolox-l"), yolox_x=factory("yolox-x"))
    head = _mod("yolox.models.yolo_head", YOLOXHead=yolox_ref.YOLOXHead)
    pafpn = _mod("yolox.models.yolo_pafpn", YOLOPAFPN=yolox_ref.YOLOPAFPN)
    utils = _mod("yolox.utils", postprocess=yolox_ref.postprocess)
    _mod("yolox", models=models, utils=utils)
    models.yolo_head, models.yolo_pafpn = head, pafpn
    # --- positional_encodings ----------------------------------------------------
    te = _mod("positional_encodings.torch_encodings",
              PositionalEncoding1D=posenc_ref.PositionalEncoding1D,
              PositionalEncoding2D=posenc_ref.PositionalEncoding2D)
    _mod("positional_encodings", torch_encodings=te)

    # --- torchvision -------------------------------------------------------------
    def box_convert(boxes, in_fmt, out_fmt):
        assert (in_fmt, out_fmt) == ("xyxy", "cxcywh")
        x1, y1, x2, y2 = boxes.unbind(-1)
        return torch.stack(((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1), -1)

    def tf_pad(img, padding, padding_mode="constant", fill=0):
        l, t, r, b = padding
        return F.pad(img, (l, r, t, b), mode=padding_mode)

    def tf_resize(img, size, antialias=True):
        return F.interpolate(img, size=size, mode="bilinear", antialias=antialias, align_corners=False)

    ops = _mod("torchvision.ops", box_convert=box_convert, nms=yolox_ref.nms)
    tff = _mod("torchvision.transforms.functional", pad=tf_pad, resize=tf_resize)
    tr = _mod("torchvision.transforms", functional=tff)
    _mod("torchvision", ops=ops, transforms=tr)

    # --- kornia ------------------------------------------------------------------
    class Boxes:
        def __init__(self, xyxy_incl):
            self.b = xyxy_incl

        @classmethod
        def from_tensor(cls, boxes, mode):
            assert mode == "xyxy_plus"
            return cls(boxes)

        def to_mask(self, height, width):
            B, nb, _ = self.b.shape
            mask = torch.zeros((B, nb, height, width))
            for i in range(B):
                for k in range(nb):
                    x1, y1, x2, y2 = (int(v) for v in self.b[i, k])
                    x1, y1 = max(x1, 0), max(y1, 0)
                    x2, y2 = min(x2 + 1, width), min(y2 + 1, height)
                    if x2 > x1 and y2 > y1:
                        mask[i, k, y1:y2, x1:x2] = 1
            return mask

    kb = _mod("kornia.geometry.boxes", Boxes=Boxes)
    kg = _mod("kornia.geometry", boxes=kb)
    ka = _mod("kornia.augmentation")
    _mod("kornia", geometry=kg, augmentation=ka)

    # --- gymnasium ---------------------------------------------------------------
    class _Space:
        def __init__(self, *a, **k):
            pass

    spaces = _mod("gymnasium.spaces", Box=_Space, Tuple=_Space, Discrete=_Space)
    _mod("gymnasium", Env=object, spaces=spaces)
    # --- never executed on the path: names only -----------------------------------
    tmd = _mod("torchmetrics.detection.mean_ap", MeanAveragePrecision=object)
    tmdet = _mod("torchmetrics.detection", mean_ap=tmd)
    _mod("torchmetrics", detection=tmdet)
    for n in ("cv2", "thop", "torchinfo", "visdom"):
        _mod(n)
    sys.modules["torchinfo"].summary = None
    sys.modules["visdom"].Visdom = object


def ref_model_config(o_cfg):
    """CfgNode the way main.args_to_config builds it (main.py:367-386)."""
    from src.utils import CfgNode
    from src.env.common import ActionInfo
    c = CfgNode()
    c.model_type, c.n_layer, c.n_head, c.n_embd = o_cfg.model_type, None, None, None
    c.embd_pdrop = c.resid_pdrop = c.attn_pdrop = 0.1
    c.image_processor, c.gpt_backbone = o_cfg.image_processor, o_cfg.gpt_backbone
    c.freeze_image_processor = False
    c.detector_conf_threshold = o_cfg.detector_conf_threshold
    c.use_pos_emb, c.no_patch_emb = o_cfg.use_pos_emb, o_cfg.no_patch_emb
    c.concat_emb, c.decoder_pos_encoding = o_cfg.concat_emb, o_cfg.decoder_pos_encoding
    c.pos_emb_size, c.dropout = o_cfg.pos_emb_size, o_cfg.dropout
    c.block_size, c.n_channels, c.patch_size = o_cfg.block_size, 3, o_cfg.patch_size
    c.image_cols = 5
    c.no_recurrent_embedding = o_cfg.no_recurrent_embedding
    c.actions_info = [ActionInfo("categorical", o_cfg.nclasses)]
    return c


def build_pair(seed, **kw):
    """Oracle model from a seed + the reference GPT holding the SAME weights."""
    from src.models.gpt import GPT
    oracle = build_gpt_ref(seed, **kw)
    ref = GPT(ref_model_config(oracle.cfg))
    missing, unexpected = ref.load_state_dict(oracle.state_dict(), strict=False)
    assert not unexpected, unexpected
    assert not missing, missing
    return oracle, ref


def to_np(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        else:
            out[k] = np.asarray(v)
    return out


def main():
    os.chdir(REF)
    sys.path.insert(0, str(REF))
    install_stubs()
    torch.set_num_threads(8)
    from src.env.general_env import NeedleGeneralEnv
    from src.env.common import Action, ACTION_DELTAS, get_actions_info
    from src.models.gpt import NewGELU, CausalSelfAttention, Block
    from src.reinforce import ReinforceTrainer
    from src.trainer import Trainer
    from src.utils import CfgNode

    # ---------------- G1: env ----------------------------------------------------
    g1 = {}
    images = torch.zeros(1, 3, 1792, 2240)
    bb = torch.tensor([[[310, 810, 400, 850], [700, 1500, 800, 1600]]])
    env = NeedleGeneralEnv(images, bb, 448, 8, 1)
    env.reset(torch.tensor([[1, 0]]))
    g1["t_env_bbox_masks"] = env.bbox_masks
    seq = [Action.RIGHT.value, Action.DOWN.value, Action.DOWN.value]
    rew, term, trunc, poss = [], [], [], []
    for a in seq:
        _, r, te, tr, info = env.step(torch.tensor([a]))
        rew.append(r); term.append(te.clone()); trunc.append(tr); poss.append(info["positions"].clone())
    g1["t_env_actions"] = torch.tensor(seq)
    g1["t_env_rewards"] = torch.stack(rew, 1)
    g1["t_env_terminated"] = torch.stack(term, 1)
    g1["t_env_truncated"] = torch.stack(trunc, 1)
    g1["t_env_positions"] = torch.stack(poss, 1)
    g1["action_deltas"] = torch.tensor([ACTION_DELTAS[Action(i)] for i in range(9)])
    # random walks with and without STOP, small images
    for tag, stop in (("nostop", False), ("stop", True)):
        B, Gh, Gw, P, T = 6, 4, 5, 16, 10
        images, bboxes, pos = synth_batch(B, Gh, Gw, P, seed=7 + stop)
        bboxes[1] = 0                                # all-padding row: marks patch (0,0)
        env = NeedleGeneralEnv(images, bboxes, P, T, 1, stop)
        p0, info = env.reset(pos.clone())
        g = torch.Generator().manual_seed(99 + stop)
        acts = torch.randint(0, 9 if stop else 8, (B, T), generator=g)
        acts[0, :] = torch.tensor([1, 1, 3, 3, 0, 0, 2, 2, 1, 3])   # deterministic wanderer
        if stop:
            acts[2, 3] = 8                            # early STOP, keeps stepping afterwards
            acts[3, 0] = 8                            # STOP at once
        rew, term, trunc, poss, vis, pat = [], [], [], [info["positions"].clone()], [], [p0]
        for t in range(T):
            p, r, te, tr, info = env.step(acts[:, t])
            rew.append(r); term.append(te.clone()); trunc.append(tr.clone())
            poss.append(info["positions"].clone()); vis.append(env.visited_patches.clone()); pat.append(p)
        g1[f"{tag}_images"] = images; g1[f"{tag}_bboxes"] = bboxes; g1[f"{tag}_start"] = pos
        g1[f"{tag}_actions"] = acts
        g1[f"{tag}_bbox_masks"] = env.bbox_masks
        g1[f"{tag}_rewards"] = torch.stack(rew, 1)
        g1[f"{tag}_terminated"] = torch.stack(term, 1)
        g1[f"{tag}_truncated"] = torch.stack(trunc, 1)
        g1[f"{tag}_positions"] = torch.stack(poss, 1)
        g1[f"{tag}_visited"] = torch.stack(vis, 1)
        g1[f"{tag}_patches"] = torch.cat(pat, 1)          # [B, T+1, 3, P, P]
        g1[f"{tag}_prop_patches_found"] = env.prop_patches_found
    g1["nclasses_stop"] = np.int64(get_actions_info(CfgNode(stop_enabled=True))[0].nclasses)
    g1["nclasses_nostop"] = np.int64(get_actions_info(CfgNode(stop_enabled=False))[0].nclasses)
    np.savez_compressed(HERE / "g1_env.npz", **to_np(g1))

    # ---------------- G7: known answers held by the reference's tests (data) -------
    env = NeedleGeneralEnv(torch.zeros((1, 3, 1792, 2240)),
                           torch.tensor([[[410, 410, 500, 500], [1500, 1500, 1600, 1600]]]), 448, 20, 1)
    tgt = env.get_detection_targets()
    expect = torch.tensor([[0, 410, 410, 447, 447], [0, 448, 410, 500, 447], [0, 410, 448, 447, 500],
                           [0, 448, 448, 500, 500], [0, 1500, 1500, 1600, 1600]])
    assert torch.equal(tgt[0], expect)               # tests/test_map.py:22-34
    patch_boxes = [[torch.tensor([[20, 40, 30, 100], [40, 60, 100, 90]]), torch.tensor([[38, 6, 90, 10]]),
                    None, torch.tensor([[70, 30, 89, 59]])]]
    offsets = torch.tensor([[[448, 0], [448, 448], [448, 896], [448, 1344]]])
    pmask = torch.tensor([[True, True, True, False]])
    full = Trainer.patch_bboxes2full_image(patch_boxes, offsets, pmask)
    np.savez_compressed(HERE / "g7_known_answers.npz", **to_np({
        "targets_bboxes": torch.tensor([[[410, 410, 500, 500], [1500, 1500, 1600, 1600]]]),
        "targets_expected": tgt[0], "p2f_box0": patch_boxes[0][0], "p2f_box1": patch_boxes[0][1],
        "p2f_box3": patch_boxes[0][3], "p2f_offsets": offsets, "p2f_masks": pmask,
        "p2f_expected": full[0]}))

    # ---------------- G2: transformer pieces ---------------------------------------
    torch.manual_seed(1234)
    cfg = CfgNode(n_embd=48, n_head=3, block_size=9, attn_pdrop=0.0, resid_pdrop=0.0)
    blk = Block(cfg).eval()
    for p in blk.parameters():
        torch.nn.init.normal_(p, 0.0, 0.2)
    x = torch.randn(2, 7, 48)
    g2 = {"gelu_in": torch.linspace(-4, 4, 33), "block_in": x, "block_out": blk(x),
          "attn_out": blk.attn(x)}
    g2["gelu_out"] = NewGELU()(g2["gelu_in"])
    for k, v in blk.state_dict().items():
        g2["sd." + k] = v
    np.savez_compressed(HERE / "g2_transformer.npz", **to_np(g2))

    # ---------------- G3: composed GPT.forward, full vs recurrent ------------------
    g3 = {}
    P, T, B = 64, 6, 3
    kw = dict(patch_size=P, block_size=T, nclasses=9, with_detector=True,
              image_processor="yolox-nano", gpt_backbone="yolox-nano")
    oracle, ref = build_pair(11, **kw)
    randomize_bn(oracle, 5); ref.load_state_dict(oracle.state_dict())
    ref.eval()
    patches, actions, positions = synth_tokens(B, T, P, 9, 5, seed=3)
    classes = torch.zeros(B, dtype=torch.long)
    with torch.no_grad():
        full_logits, full_emb = ref(patches, actions, classes, positions)
        emb, rec_logits = None, []
        for t in range(T):
            lg, emb = ref(patches[:, :t + 1], actions[:, :t + 1], classes, positions[:, :t + 1], emb)
            rec_logits.append(lg[:, -1])
        fpn = ref.gpt_backbone(patches[:, 0])
        patch_emb = ref.embed_patches(patches[:, :2])
    g3.update(seed=np.int64(11), bn_seed=np.int64(5), tok_seed=np.int64(3), actions=actions,
              positions=positions, full_logits=full_logits, full_emb=full_emb,
              rec_logits=torch.stack(rec_logits, 1), rec_emb=emb,
              fpn0=fpn[0], fpn1=fpn[1], fpn2=fpn[2], patch_emb=patch_emb)
    np.savez_compressed(HERE / "g3_gpt_forward.npz", **to_np(g3))

    # ---------------- G4 + G5: reference rollout (greedy) and metrics --------------
    g4 = {}
    shell = object.__new__(ReinforceTrainer)
    shell.device = "cpu"
    shell.model = ref
    shell.yolox_model = lambda: ref.yolox
    shell.config = CfgNode(reward_norm=True)
    shell.entropy_weight = 0.01
    shell.stop_enabled = True
    shell.last_return_values, shell.last_return_mean, shell.last_return_std = [], 0, 1
    B, Gh, Gw = 4, 4, 5
    images, bboxes, pos = synth_batch(B, Gh, Gw, P, seed=21)
    env = NeedleGeneralEnv(images, bboxes, P, T, 1, True)
    torch.manual_seed(77)                            # env.reset() draws from the CPU generator
    with torch.no_grad():
        ro = shell.rollout(env, do_detection=False, sample_actions=False)
    g4.update(batch_seed=np.int64(21), bboxes=bboxes, start=ro["positions"][:, 0], T=np.int64(T), P=np.int64(P))
    for k in ("rewards", "returns", "logprobs", "entropies", "masks", "logit_masks", "positions"):
        g4[k] = ro[k]
    m1 = shell.compute_metrics(ro)
    shell._compute_last_returns_mean_std()
    m2 = shell.compute_metrics(ro)                   # second window: normalised advantages
    for tag, m in (("m1", m1), ("m2", m2)):
        for k, v in m.items():
            g4[f"{tag}.{k}"] = v
    g4["norm_mean"], g4["norm_std"] = shell.last_return_mean, shell.last_return_std
    np.savez_compressed(HERE / "g4_rollout.npz", **to_np(g4))
    for f in sorted(HERE.glob("*.npz")):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
