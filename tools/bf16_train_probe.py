#!/usr/bin/env python3
"""Evidence for DESIGN.md §6 "bf16 training": the supervised step of BASELINE configs[1] (B=4, T=8, 448 px) with bf16
activation storage (JN_ALLOW_BF16_TRAIN=1 lifts the engine's refusal) against the fp32 engine and the fp32 CPU oracle, on
(a) the worst-conditioned input there is — i.i.d. noise patches through a random-init net (near-constant deep maps) — and
(b) structured patches (smooth blobs + edges, the statistics of real imagery: the maps keep their variance through the
depth).  Prints per-family gradient distances (relative L2 and max-norm), logits and loss distances.
usage (GPU box): JN_ALLOW_BF16_TRAIN=1 python3 tools/bf16_train_probe.py > gpurun_out/bf16_train_probe.txt"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("JN_ALLOW_BF16_TRAIN", "1")

import jolineedle_amd as ja
from tests.helpers import make_pair, synth_tokens


def structured(B, T, P, seed):
    """Smooth random fields + a few hard edges per patch, in [0, 1]."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand((B * T, 3, P // 32, P // 32), generator=g)
    x = torch.nn.functional.interpolate(low, size=(P, P), mode="bicubic", align_corners=False)
    mid = torch.rand((B * T, 3, P // 8, P // 8), generator=g)
    x = 0.7 * x + 0.3 * torch.nn.functional.interpolate(mid, size=(P, P), mode="bilinear", align_corners=False)
    for i in range(B * T):
        for _ in range(3):
            x0, y0 = (int(torch.randint(0, P - 64, (1,), generator=g)) for _ in range(2))
            w, h = (int(torch.randint(16, 64, (1,), generator=g)) for _ in range(2))
            x[i, :, y0:y0 + h, x0:x0 + w] = torch.rand((3, 1, 1), generator=g)
    return x.clamp(0, 1).view(B, T, 3, P, P)


def run(kind):
    B, T, P = 4, 8, 448
    kw = dict(patch_size=P, block_size=T, with_detector=False, image_processor=None, max_batch=B * T)
    p32, oracle = make_pair(13, **kw)
    p16, _ = make_pair(13, act_dtype="bf16", **kw)
    patches, cur, positions = synth_tokens(B, T, P, 9, 5, seed=21)
    if kind == "structured":
        patches = structured(B, T, P, 5)
    nxt = torch.randint(0, 9, (B, T), generator=torch.Generator().manual_seed(4))
    masks = torch.ones((B, T), dtype=torch.long)
    oracle.train(); oracle.zero_grad()
    lg, _ = oracle(patches, cur, torch.zeros(B, dtype=torch.long), positions)
    loss = torch.nn.functional.cross_entropy(lg.reshape(B * T, 9), nxt.flatten())
    loss.backward()
    ref = {n: p.grad.detach().clone() for n, p in oracle.named_parameters() if p.grad is not None}
    cfg = ja.CfgNode(stop_enabled=True, stop_weight=1.0, learning_rate=1e-3, gradient_accumulation=1)
    out = {}
    for tag, prod in (("fp32 engine", p32), ("bf16 engine", p16)):
        m = ja.SupervisedTrainer(cfg, prod).train_step(patches, cur, nxt, positions, masks, optimizer_step=False)
        grads = prod.engine_grads()
        fam = {}
        for n, r in ref.items():
            if r.abs().max() < 1e-12 or n not in grads:
                continue
            f = "encoder first stages" if n.startswith(("gpt_backbone.backbone.stem", "gpt_backbone.backbone.dark2", "gpt_backbone.backbone.dark3")) \
                else "encoder" if n.startswith("gpt_backbone") else "embed_fpn" if n.startswith("embed_fpn") else "decision"
            d = grads[n].double() - r.double()
            l2 = float(d.norm() / r.double().norm())
            mx = float(d.abs().max() / r.abs().max())
            cur_ = fam.setdefault(f, [0.0, 0.0, 0])
            cur_[0] = max(cur_[0], l2); cur_[1] = max(cur_[1], mx); cur_[2] += 1
        out[tag] = (float((m["logits"].cpu() - lg.detach()).abs().max()), abs(float(m["loss"]) - float(loss)), fam)
    print(f"== {kind} patches: supervised step B={B} T={T} P={P}, distances from the fp32 CPU oracle (worst tensor per family)")
    for tag, (dl, dloss, fam) in out.items():
        print(f"  {tag}: logits max |d| {dl:.2e}   loss |d| {dloss:.2e}")
        for f, (l2, mx, n) in fam.items():
            print(f"      {f:22s} {n:4d} tensors   relative L2 {l2:.2e}   max-norm {mx:.2e}")


if __name__ == "__main__":
    assert torch.cuda.is_available()
    for kind in ("noise", "structured"):
        run(kind)
