"""Oracle restatement of the JoliNeedle decision model (GPT + ActionHead).

TEST INFRASTRUCTURE ONLY.  Follows src/models/gpt.py:31-140 (NewGELU,
CausalSelfAttention, Block), :162-329 (construction / init), :331-534
(embeddings, recurrence, forward) and src/models/action_head.py:14-33.
Pinned against the reference's own classes by tests/golden/make_golden.py
(G2/G3 vectors).  Submodule names are the reference's, so ``state_dict()``
round-trips with reference checkpoints.
"""
import math
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .posenc_ref import PositionalEncoding1D, PositionalEncoding2D
from .yolox_ref import NeedleYOLOXRef, build_head, build_pafpn

# src/models/gpt.py:190-218
GPT_ZOO = {
    "openai-gpt": (12, 12, 768), "gpt2": (12, 12, 768), "gpt2-medium": (24, 16, 1024),
    "gpt2-large": (36, 20, 1280), "gpt2-xl": (48, 25, 1600), "gopher-44m": (8, 16, 512),
    "gpt-mini": (6, 6, 192), "gpt-micro": (4, 4, 128), "gpt-nano": (3, 3, 48),
    "gpt-pico": (2, 2, 32),
}


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    """NewGELU, gpt.py:37-47."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3.0))))


class _Attn(nn.Module):
    def __init__(self, n_embd, n_head, block_size):
        super().__init__()
        self.c_attn = nn.Linear(n_embd, 3 * n_embd)
        self.c_proj = nn.Linear(n_embd, n_embd)
        self.register_buffer(
            "bias", torch.tril(torch.ones(block_size, block_size)).view(1, 1, block_size, block_size))
        self.n_head, self.n_embd = n_head, n_embd
        self.drop, self.layer = None, 0          # injected dropout masks (oracle/dropout_ref.py), None = no dropout

    def forward(self, x):
        B, T, C = x.shape
        hs = C // self.n_head
        q, k, v = (t.view(B, T, self.n_head, hs).transpose(1, 2)
                   for t in self.c_attn(x).split(C, dim=2))
        att = (q @ k.transpose(-2, -1)) / math.sqrt(hs)
        att = att.masked_fill(self.bias[:, :, :T, :T] == 0, float("-inf")).softmax(dim=-1)
        if self.drop is not None and self.training:
            att = att * self.drop.att(B, self.n_head, T, self.layer, att.dtype)           # attn_dropout, gpt.py:100
        y = self.c_proj((att @ v).transpose(1, 2).reshape(B, T, C))
        if self.drop is not None and self.training:
            y = y * self.drop.vec(B, T, C, self.layer, 2, y.dtype)                         # resid_dropout, gpt.py:107
        return y


class _Block(nn.Module):
    def __init__(self, n_embd, n_head, block_size):
        super().__init__()
        self.ln_1 = nn.LayerNorm(n_embd)
        self.attn = _Attn(n_embd, n_head, block_size)
        self.ln_2 = nn.LayerNorm(n_embd)
        self.mlp = nn.ModuleDict(dict(c_fc=nn.Linear(n_embd, 4 * n_embd),
                                      c_proj=nn.Linear(4 * n_embd, n_embd)))
        self.drop, self.layer = None, 0

    def forward(self, x):
        x = x + self.attn(self.ln_1(x))
        m = self.mlp["c_proj"](gelu_tanh(self.mlp["c_fc"](self.ln_2(x))))
        if self.drop is not None and self.training:
            m = m * self.drop.vec(x.shape[0], x.shape[1], x.shape[2], self.layer, 3, m.dtype)   # mlp dropout, gpt.py:124
        return x + m


class _ActionHead(nn.Module):
    def __init__(self, nclasses_list, n_embd):
        super().__init__()
        self.lm_heads = nn.ModuleList(nn.Linear(n_embd, n, bias=False) for n in nclasses_list)

    def forward(self, x):
        if len(self.lm_heads) == 1:
            return self.lm_heads[0](x)
        return torch.stack([h(x) for h in self.lm_heads], dim=2)


def default_model_config(**kw) -> SimpleNamespace:
    """The README / tests recipe (README.md:56-131, tests/test_rl.py:13-46):
    gpt-nano + yolox-nano backbone + yolox-s detector, concat embeddings,
    sinusoid decoder positions, 2-D patch positions, dropout 0, STOP enabled."""
    cfg = dict(model_type="gpt-nano", image_processor="yolox-s", gpt_backbone="yolox-nano",
               patch_size=448, n_channels=3, block_size=20, nclasses=9,
               use_pos_emb=True, no_patch_emb=False, concat_emb=True,
               decoder_pos_encoding=True, pos_emb_size=25, dropout=0.0,
               detector_conf_threshold=0.5, no_recurrent_embedding=False,
               with_detector=True)
    cfg.update(kw)
    return SimpleNamespace(**cfg)


class GPTRef(nn.Module):
    """Only the configuration the hot path uses is restated: single categorical
    action, ``concat_emb`` or mean merge, optional patch / position embeddings,
    optional standalone ``gpt_backbone``.  Dropout: masks injected by ``enable_dropout`` (the engine's Philox masks)."""

    def __init__(self, cfg):
        super().__init__()
        # cfg.dropout is carried for the config surface only: the oracle drops with INJECTED masks (enable_dropout)
        self.cfg = cfg
        n_layer, n_head, C = GPT_ZOO[cfg.model_type]
        self.n_embd, self.n_head, self.n_layer = C, n_head, n_layer
        self.block_size = cfg.block_size
        self.patch_size = cfg.patch_size
        self.token_offset = 1
        self.action_head = _ActionHead([cfg.nclasses], C)
        self.positional_encoding = PositionalEncoding2D(C)
        if cfg.decoder_pos_encoding:
            self.decoder_token_pos_enc = PositionalEncoding1D(C)
        self.embed_class = nn.Embedding(100, C)
        if cfg.concat_emb:
            n_emb = 2 + (0 if cfg.no_patch_emb else 1) + (1 if cfg.use_pos_emb else 0)
            self.project_concat = nn.Linear(n_emb * C, C)
        if cfg.with_detector:
            self.yolox = NeedleYOLOXRef(build_pafpn(cfg.image_processor),
                                        build_head(cfg.image_processor, 1),
                                        cfg.detector_conf_threshold)
        if cfg.gpt_backbone:
            self.gpt_backbone = build_pafpn(cfg.gpt_backbone)
        if not cfg.no_patch_emb:
            enc = self.gpt_backbone if cfg.gpt_backbone else self.yolox.backbone
            with torch.no_grad():
                was = enc.training
                enc.eval()
                last = enc(torch.zeros(1, cfg.n_channels, cfg.patch_size, cfg.patch_size))[-1]
                enc.train(was)
            self.embed_fpn = nn.Sequential(
                nn.Conv2d(last.shape[1], C, 1, 1, 0, bias=False), nn.ReLU(),
                nn.Flatten(start_dim=1), nn.Linear(last.shape[2] * last.shape[3] * C, C))
        self.transformer = nn.ModuleDict(dict(
            wte=nn.Embedding(cfg.nclasses, C),
            wpe=nn.Embedding(cfg.pos_emb_size, C),
            h=nn.ModuleList([_Block(C, n_head, cfg.block_size + 1) for _ in range(n_layer)]),
            ln_f=nn.LayerNorm(C)))
        # gpt.py:323-329, 536-545: only Linear / Embedding / LayerNorm are re-initialised.
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0.0, 0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, 0.0, 0.02)
        for name, p in self.named_parameters():
            if name.endswith("c_proj.weight"):
                nn.init.normal_(p, 0.0, 0.02 / math.sqrt(2 * n_layer))

    # -- embeddings (gpt.py:331-479) -------------------------------------------------
    def token_positions(self, tok_emb):
        B, T, _ = tok_emb.shape
        if self.cfg.decoder_pos_encoding:
            return self.decoder_token_pos_enc(tok_emb)
        return self.transformer["wpe"](torch.arange(T)).unsqueeze(0).expand(B, T, -1)

    def embed_patches(self, patches):
        B, T = patches.shape[:2]
        flat = patches.reshape(B * T, *patches.shape[2:])
        if self.cfg.gpt_backbone:
            last = self.gpt_backbone(flat)[-1]
        else:
            last = self.yolox.backbone(flat)[-1].detach()
        return self.embed_fpn(last).view(B, T, -1)

    def embed_patch_position(self, positions):
        rows, cols = positions[..., 0], positions[..., 1]
        table = self.positional_encoding(
            torch.zeros(1, int(cols.max()) + 1, int(rows.max()) + 1, self.n_embd))[0]
        return table[cols, rows]

    def embed_inputs(self, patches, actions, classes, positions, prev_embeddings):
        if self.cfg.no_recurrent_embedding:
            prev_embeddings = None
        if prev_embeddings is not None:           # recurrent mode: newest token only
            actions, patches = actions[:, -1:], patches[:, -1:]
            if self.cfg.use_pos_emb:
                positions = positions[:, -1:]
        tok = self.transformer["wte"](actions)
        parts = [tok, self.token_positions(tok)]
        if not self.cfg.no_patch_emb:
            parts.append(self.embed_patches(patches))
        if self.cfg.use_pos_emb:
            parts.append(self.embed_patch_position(positions))
        if self.cfg.concat_emb:
            emb = self.project_concat(torch.cat(parts, dim=2))
        else:
            emb = torch.stack(parts, dim=2).mean(dim=2)
        if prev_embeddings is not None:
            return torch.cat((prev_embeddings, emb), dim=1)
        return torch.cat((self.embed_class(classes).unsqueeze(1), emb), dim=1)

    def enable_dropout(self, p: float, seed: int):
        """Dropout with the engine's masks injected (oracle/dropout_ref.py); train mode only; p = 0 switches it off."""
        from .dropout_ref import DropoutMasks
        self.drop = DropoutMasks(p, seed, self.block_size + 1) if p > 0 else None
        for l, blk in enumerate(self.transformer["h"]):
            blk.drop = blk.attn.drop = self.drop
            blk.layer = blk.attn.layer = l

    def decode(self, final_emb):
        x = final_emb
        if getattr(self, "drop", None) is not None and self.training:
            x = x * self.drop.vec(x.shape[0], x.shape[1], x.shape[2], 0, 0, x.dtype)      # transformer.drop, gpt.py:525
        for blk in self.transformer["h"]:
            x = blk(x)
        return self.transformer["ln_f"](x)

    def forward(self, patches, actions, classes, positions=None, prev_embeddings=None):
        assert actions.shape[1] <= self.block_size
        final_emb = self.embed_inputs(patches, actions, classes, positions, prev_embeddings)
        logits = self.action_head(self.decode(final_emb))[:, self.token_offset:].contiguous()
        return logits, final_emb


def build_gpt_ref(seed: int = 0, **cfg_kw) -> GPTRef:
    """Deterministic random-init model (no pretrained weights offline)."""
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    model = GPTRef(default_model_config(**cfg_kw))
    torch.random.set_rng_state(gen_state)
    return model
