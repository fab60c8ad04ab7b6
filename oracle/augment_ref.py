"""CPU restatement (test infrastructure only) of the detection-augmentation arithmetic of
jolineedle_amd/augment.py / csrc/kernels_aug.hip: the op chain of Trainer.init_detection (src/trainer.py:176-186)
for GIVEN per-patch parameters — colour gains + clamp, grayscale (0.299, 0.587, 0.114), 3x3 Gaussian with reflect
padding, plasma shadow (the counter-based fractal restated in numpy), additive noise, 3x3 motion kernel with zero padding — written with torch's own conv / pad ops.
kornia (the reference's provider of these ops) is not in the reference tree: parity unpinned."""
import numpy as np
import torch
import torch.nn.functional as F

PLASMA_OCTAVES = 7


def _fmix32(v):
    v = v.astype(np.uint32)
    with np.errstate(over="ignore"):
        v ^= v >> np.uint32(16); v *= np.uint32(0x85EBCA6B)
        v ^= v >> np.uint32(13); v *= np.uint32(0xC2B2AE35)
        v ^= v >> np.uint32(16)
    return v


def plasma_ref(seed: int, n: int, P: int, rough: float, stretch: float) -> np.ndarray:
    """The plasma-shadow fractal of patch n ([P, P], float32 in [0, 1]): csrc/kernels_aug.hip aug_plasma, same hash, same
    fp32 operation order."""
    f32 = np.float32
    with np.errstate(over="ignore"):
        pseed = _fmix32(np.array([(seed & 0xFFFFFFFF) ^ ((seed >> 32) & 0xFFFFFFFF)], np.uint32) ^
                        (np.uint32(n) * np.uint32(0x9E3779B9) + np.uint32(0x7F4A7C15)))[0]
    ys, xs = np.meshgrid(np.arange(P, dtype=np.int64), np.arange(P, dtype=np.int64), indexing="ij")

    def lattice(o, iy, ix):
        with np.errstate(over="ignore"):
            h = iy.astype(np.uint32) * np.uint32(0x85EBCA77) ^ ix.astype(np.uint32) * np.uint32(0x9E3779B1) ^ np.uint32((o + 1) * 0xC2B2AE3D & 0xFFFFFFFF)
            h = _fmix32(_fmix32(h) ^ pseed)
        return (h >> np.uint32(8)).astype(f32) * f32(1.0 / 16777216.0)
    f = np.zeros((P, P), f32); wsum = f32(0.0); wgt = f32(1.0)
    for o in range(PLASMA_OCTAVES):
        cell = f32(P) / f32(2 << o)
        fy, fx = ys.astype(f32) / cell, xs.astype(f32) / cell
        iy, ix = fy.astype(np.int32), fx.astype(np.int32)
        ty, tx = fy - iy.astype(f32), fx - ix.astype(f32)
        a, b, c, d = lattice(o, iy, ix), lattice(o, iy, ix + 1), lattice(o, iy + 1, ix), lattice(o, iy + 1, ix + 1)
        top, bot = a + (b - a) * tx, c + (d - c) * tx
        f = f + wgt * (top + (bot - top) * ty)
        wsum = f32(wsum + wgt)
        wgt = f32(wgt * f32(rough))
    f = f32(0.5) + (f / wsum - f32(0.5)) * f32(stretch)
    return np.clip(f, 0.0, 1.0).astype(f32)


def augment_ref(x: torch.Tensor, params: torch.Tensor, noise: torch.Tensor, seed: int = 0) -> torch.Tensor:
    x = x.clone().float()
    n = x.shape[0]
    out = torch.empty_like(x)
    for i in range(n):
        p = params[i]
        img = x[i]
        if float(p[0]) != 1.0 or float(p[1]) != 1.0:
            img = torch.stack((img[0] * p[0], img[1], img[2] * p[1])).clamp(0.0, 1.0)
        if float(p[2]) != 0.0:
            l = 0.299 * img[0] + 0.587 * img[1] + 0.114 * img[2]
            img = torch.stack((l, l, l))
        k1 = torch.stack((p[4], p[3], p[4]))
        k2 = torch.outer(k1, k1)[None, None].repeat(3, 1, 1, 1)
        img = F.conv2d(F.pad(img[None], (1, 1, 1, 1), mode="reflect"), k2, groups=3)[0]
        if params.shape[1] > 18 and float(p[15]) != 0.0:           # plasma shadow, between the blur and the noise
            f = torch.from_numpy(plasma_ref(seed, i, x.shape[2], float(p[17]), float(p[18])))
            img = img * torch.where(f < p[16], 1.0 + p[15], torch.ones(()))
        img = img + p[5] * noise[i]
        km = p[6:15].reshape(1, 1, 3, 3).repeat(3, 1, 1, 1)
        out[i] = F.conv2d(img[None], km, padding=1, groups=3)[0]
    return out
