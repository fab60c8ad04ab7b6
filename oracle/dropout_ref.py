"""Oracle side of the decision transformer's dropout (TEST INFRASTRUCTURE ONLY).

The reference drops with torch's generator (src/models/gpt.py:65-66, 100, 107, 124, 314, 525); a device engine cannot
reproduce that stream, so parity is defined with the MASK INJECTED: the engine's keep mask is a pure function
(Philox4x32-10, include/jnroll.h jn_set_dropout, csrc/jn_device.h drop_scale) of (seed, agent, token, layer, site, index),
restated here in numpy, and `GPTRef.enable_dropout` multiplies the same masks in at the reference's four dropout sites.
Philox4x32-10 itself is pinned by the known-answer vectors of Random123 (tests/test_oracle_golden.py).
"""
import numpy as np
import torch

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32(seed: int, c0, c1, c2, c3):
    """Vectorised Philox4x32-10: counters are uint32 arrays of one shape, the key is the 64-bit seed; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & MASK32).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & MASK32).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return c0, c1, c2, c3


def drop_scale(seed: int, b, tok, layer: int, site: int, idx, p: float) -> np.ndarray:
    """Keep-scale (0 or 1 / (1 - p)) — csrc/jn_device.h drop_scale, vectorised over broadcastable b / tok / idx."""
    b, tok, idx = np.broadcast_arrays(np.asarray(b), np.asarray(tok), np.asarray(idx))
    tag = np.uint32(0x44520000 | (layer << 4) | site)
    r = philox4x32(seed, b.astype(np.uint32), tok.astype(np.uint32), np.full(b.shape, tag, np.uint32), (idx >> 2).astype(np.uint32))
    k = idx & 3
    w = np.where(k == 0, r[0], np.where(k == 1, r[1], np.where(k == 2, r[2], r[3])))
    u = (w >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return np.where(u >= np.float32(p), np.float32(1.0 / (1.0 - p)), np.float32(0.0)).astype(np.float32)


class DropoutMasks:
    """Mask provider for GPTRef: `tok0` = absolute index of the first token of the tensor being masked."""

    def __init__(self, p: float, seed: int, tmax: int):
        self.p, self.seed, self.tmax = float(p), int(seed), int(tmax)

    def vec(self, B: int, T: int, C: int, layer: int, site: int, dtype, tok0: int = 0) -> torch.Tensor:
        b = np.arange(B)[:, None, None]
        t = tok0 + np.arange(T)[None, :, None]
        i = np.arange(C)[None, None, :]
        return torch.from_numpy(drop_scale(self.seed, b, t, layer, site, i, self.p)).to(dtype)

    def att(self, B: int, nh: int, T: int, layer: int, dtype) -> torch.Tensor:
        """[B, nh, T(query token), T(key)]: idx = head * Tmax + key, keyed by the query token."""
        b = np.arange(B)[:, None, None, None]
        h = np.arange(nh)[None, :, None, None]
        q = np.arange(T)[None, None, :, None]
        k = np.arange(T)[None, None, None, :]
        return torch.from_numpy(drop_scale(self.seed, b, q, layer, 1, h * self.tmax + k, self.p)).to(dtype)
