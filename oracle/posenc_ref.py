"""Oracle restatement of ``positional_encodings.torch_encodings`` (>=6.0.1).

TEST INFRASTRUCTURE ONLY.  The package is a third-party dependency
(requirements.txt:11) absent from /root/reference: **parity unpinned**.
Restated from the published v6 algorithm (interleaved sin/cos), anchored on the
reference call sites src/models/gpt.py:223-225, 345, 397-417.
"""
import math

import torch
import torch.nn as nn


def _interleave_sin_cos(angles: torch.Tensor) -> torch.Tensor:
    """[..., F] -> [..., 2F] as (sin f0, cos f0, sin f1, cos f1, ...)."""
    return torch.stack((angles.sin(), angles.cos()), dim=-1).flatten(-2, -1)


class PositionalEncoding1D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.org_channels = channels
        channels = int(math.ceil(channels / 2) * 2)
        self.channels = channels
        inv_freq = 1.0 / (10000 ** (torch.arange(0, channels, 2).float() / channels))
        self.register_buffer("inv_freq", inv_freq)

    def forward(self, tensor: torch.Tensor) -> torch.Tensor:
        b, n, c = tensor.shape
        pos = torch.arange(n, dtype=self.inv_freq.dtype)
        emb = _interleave_sin_cos(torch.outer(pos, self.inv_freq))      # [n, channels]
        return emb[None, :, :c].repeat(b, 1, 1)


class PositionalEncoding2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.org_channels = channels
        channels = int(math.ceil(channels / 4) * 2)
        self.channels = channels
        inv_freq = 1.0 / (10000 ** (torch.arange(0, channels, 2).float() / channels))
        self.register_buffer("inv_freq", inv_freq)

    def forward(self, tensor: torch.Tensor) -> torch.Tensor:
        b, nx, ny, c = tensor.shape
        ex = _interleave_sin_cos(torch.outer(torch.arange(nx, dtype=self.inv_freq.dtype), self.inv_freq))
        ey = _interleave_sin_cos(torch.outer(torch.arange(ny, dtype=self.inv_freq.dtype), self.inv_freq))
        emb = torch.zeros((nx, ny, 2 * self.channels), dtype=tensor.dtype)
        emb[:, :, : self.channels] = ex[:, None, :]
        emb[:, :, self.channels:] = ey[None, :, :]
        return emb[None, :, :, :c].repeat(b, 1, 1, 1)


def posenc2d_row_col(rows: torch.Tensor, cols: torch.Tensor, n_embd: int) -> torch.Tensor:
    """Closed form of GPT.embed_patch_position (src/models/gpt.py:386-417): the first
    ``ch`` channels encode the COLUMN, the next ``ch`` the ROW, truncated to n_embd."""
    ch = int(math.ceil(n_embd / 4) * 2)
    inv_freq = 1.0 / (10000 ** (torch.arange(0, ch, 2).float() / ch))
    ex = _interleave_sin_cos(cols.float()[..., None] * inv_freq)
    ey = _interleave_sin_cos(rows.float()[..., None] * inv_freq)
    return torch.cat((ex, ey), dim=-1)[..., :n_embd]
