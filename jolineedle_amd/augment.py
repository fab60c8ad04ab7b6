"""On-device detection augmentation: ``Trainer.init_detection`` of the reference
(src/trainer.py:176-186) builds a kornia ``nn.Sequential`` — RandomPlanckianJitter("CIED"), RandomGrayscale(p=0.2),
RandomGaussianBlur((3, 3), sigma (0.1, 2.0)), RandomPlasmaShadow, RandomGaussianNoise(std 0.05, p=0.5),
RandomMotionBlur(3, (-180, 180), 0.0, p=0.3) — and applies it to the detector patches (src/reinforce.py:332-333) and,
in supervised mode, to the trajectory patches as well (src/supervised.py:855-861, 884-885).

Here the per-patch random parameters are drawn on the host (a few floats per patch) and the whole chain runs as ONE
fused HIP pass (``jn_augment_patches``): one read and one write of every element instead of one round trip per op.

kornia is not part of the reference tree, so its parameter conventions are restated from its documentation
(parity unpinned, oracle/augment_ref.py is the checker of the arithmetic, not of kornia):
* PlanckianJitter multiplies red and blue by a (r/g, b/g) pair drawn from a table of illuminants and clamps to [0, 1];
  the CIED table is regenerated from the CIE D-series formulas (``planckian_cied_table``; kornia's own numbers are data of
  kornia) and can be overridden with ``planckian_coeffs`` [K, 2].
* PlasmaShadow multiplies the (blurred) patch by 1 + shade_intensity wherever a per-patch fractal falls below
  shade_quantity.  kornia's recursive diamond-square generator is restated as counter-based fractional-Brownian value
  noise with the same octave structure (csrc/kernels_aug.hip ``aug_plasma``), which every pixel evaluates independently.
"""
import math
from typing import Optional

import torch

from . import _lib
from ._lib import check, ptr

NPARAM = 20
PLASMA_OCTAVES = 7          # csrc/kernels_aug.hip AUG_OCT


def planckian_cied_table() -> torch.Tensor:
    """(r/g, b/g) white-balance gains of the CIE D-series daylight illuminants 4000 K .. 15000 K in 500 K steps: the
    illuminant family of kornia's RandomPlanckianJitter(mode="CIED") (src/trainer.py:177).  kornia's own table is data
    of a package that is not in the reference tree, so it is regenerated here from the published CIE formulas
    (chromaticity x_D(T), y_D = -3 x_D^2 + 2.87 x_D - 0.275; XYZ -> linear sRGB, D65 matrix), normalised to green."""
    rows = []
    for T in range(4000, 15001, 500):
        t = float(T)
        if t <= 7000.0:
            x = -4.6070e9 / t ** 3 + 2.9678e6 / t ** 2 + 0.09911e3 / t + 0.244063
        else:
            x = -2.0064e9 / t ** 3 + 1.9018e6 / t ** 2 + 0.24748e3 / t + 0.237040
        y = -3.0 * x * x + 2.870 * x - 0.275
        X, Y, Z = x / y, 1.0, (1.0 - x - y) / y
        r = 3.2406 * X - 1.5372 * Y - 0.4986 * Z
        g = -0.9689 * X + 1.8758 * Y + 0.0415 * Z
        b = 0.0557 * X - 0.2040 * Y + 1.0570 * Z
        rows.append((r / g, b / g))
    return torch.tensor(rows, dtype=torch.float32)


def plasma_stretch(roughness: torch.Tensor) -> torch.Tensor:
    """1 / (5 sigma) of the value-noise fractal (octave weights roughness^o): stands in for the per-sample min-max
    normalisation of kornia's plasma map.  A bilinearly interpolated uniform lattice has variance ~ (1/12) (2/3)^2."""
    o = torch.arange(PLASMA_OCTAVES, dtype=torch.float32)
    w = roughness[:, None] ** o[None, :]
    sigma = torch.sqrt((w * w).sum(1) * (1.0 / 12.0) * (4.0 / 9.0)) / w.sum(1)
    return 1.0 / (5.0 * sigma)


def gaussian_weights3(sigma: torch.Tensor):
    """(centre, side) weights of the normalised 3-tap Gaussian exp(-x^2 / 2 sigma^2), x in {-1, 0, 1}."""
    side = torch.exp(-0.5 / (sigma * sigma))
    norm = 1.0 + 2.0 * side
    return 1.0 / norm, side / norm


def motion_kernel3(angle_deg: float) -> torch.Tensor:
    """3x3 motion-blur kernel: the middle row [1/2, 1/2, 1/2] (direction 0) rotated by `angle_deg` about the centre with
    bilinear sampling (zeros outside), normalised to sum 1."""
    th = math.radians(angle_deg)
    c, s = math.cos(th), math.sin(th)
    line = torch.zeros(3, 3)
    line[1, :] = 0.5
    k = torch.zeros(3, 3)
    for y in range(3):
        for x in range(3):
            # source coordinate of output (x, y): rotate back about the centre (1, 1)
            sx = c * (x - 1) + s * (y - 1) + 1
            sy = -s * (x - 1) + c * (y - 1) + 1
            x0, y0 = math.floor(sx), math.floor(sy)
            v = 0.0
            for dy in (0, 1):
                for dx in (0, 1):
                    xi, yi = x0 + dx, y0 + dy
                    if 0 <= xi < 3 and 0 <= yi < 3:
                        v += float(line[yi, xi]) * (1 - abs(sx - xi)) * (1 - abs(sy - yi))
            k[y, x] = v
    return k / k.sum()


class DetectionAugment:
    """Callable with the reference's usage: ``patches = self.detection_augment(patches)`` on [N, 3, P, P] device tensors."""

    def __init__(self, planckian_coeffs: Optional[torch.Tensor] = None, p_planckian: float = 0.5, p_gray: float = 0.2,
                 p_blur: float = 0.5, sigma=(0.1, 2.0), p_noise: float = 0.5, noise_std: float = 0.05,
                 p_motion: float = 0.3, angle=(-180.0, 180.0), p_shadow: float = 0.5, shade_intensity=(-0.2, 0.0),
                 shade_quantity=(0.0, 0.4), roughness=(0.1, 0.7), seed: Optional[int] = None):
        # the CIED table is embedded (planckian_cied_table); pass another [K, 2] table of (r/g, b/g) gains to override it
        self.planckian = planckian_cied_table() if planckian_coeffs is None else planckian_coeffs
        self.p = (p_planckian, p_gray, p_blur, p_noise, p_motion)
        self.p_shadow, self.shade_intensity, self.shade_quantity, self.roughness = p_shadow, shade_intensity, shade_quantity, roughness
        self.sigma, self.noise_std, self.angle = sigma, noise_std, angle
        self.gen = torch.Generator()
        if seed is not None:
            self.gen.manual_seed(seed)
        self.calls = 0

    def sample_params(self, n: int) -> torch.Tensor:
        """[n, NPARAM] = r_gain, b_gain, gray, w0, w1, noise_std, k[9], shade intensity, shade quantity, roughness, fractal
        stretch, pad — identity where an op was not drawn."""
        g = self.gen
        u = torch.rand((n, 5), generator=g)
        prm = torch.zeros((n, NPARAM))
        prm[:, 0] = prm[:, 1] = prm[:, 3] = 1.0
        prm[:, 10] = 1.0                                               # delta motion kernel: k[1][1]
        if self.planckian is not None:
            idx = torch.randint(0, self.planckian.shape[0], (n,), generator=g)
            on = u[:, 0] < self.p[0]
            prm[on, 0] = self.planckian[idx[on], 0].float()
            prm[on, 1] = self.planckian[idx[on], 1].float()
        prm[:, 2] = (u[:, 1] < self.p[1]).float()
        sig = self.sigma[0] + (self.sigma[1] - self.sigma[0]) * torch.rand(n, generator=g)
        w0, w1 = gaussian_weights3(sig)
        on = u[:, 2] < self.p[2]
        prm[on, 3], prm[on, 4] = w0[on], w1[on]
        prm[:, 5] = (u[:, 3] < self.p[3]).float() * self.noise_std
        ang = self.angle[0] + (self.angle[1] - self.angle[0]) * torch.rand(n, generator=g)
        for i in torch.nonzero(u[:, 4] < self.p[4]).flatten().tolist():
            prm[i, 6:15] = motion_kernel3(float(ang[i])).flatten()
        # RandomPlasmaShadow(shade_intensity=(-0.2, 0), shade_quantity=(0, 0.4), p=0.5), roughness kornia's default (0.1, 0.7)
        us = torch.rand((n, 4), generator=g)
        on = us[:, 0] < self.p_shadow
        lerp = lambda r, t: r[0] + (r[1] - r[0]) * t
        rough = lerp(self.roughness, us[:, 3])
        prm[on, 15] = lerp(self.shade_intensity, us[on, 1])
        prm[on, 16] = lerp(self.shade_quantity, us[on, 2])
        prm[on, 17] = rough[on]
        prm[on, 18] = plasma_stretch(rough)[on]
        return prm

    def __call__(self, patches: torch.Tensor, params: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert patches.is_cuda and patches.dim() == 4 and patches.shape[1] == 3 and patches.shape[2] == patches.shape[3], \
            "patches must be a [N, 3, P, P] device tensor"
        x = patches.to(torch.float32).contiguous()
        n, P = x.shape[0], x.shape[2]
        if params is None:
            params = self.sample_params(n)
        prm = params.to(x.device, torch.float32).contiguous()
        assert prm.shape == (n, NPARAM)
        nz = None if noise is None else noise.to(x.device, torch.float32).contiguous()
        out = torch.empty_like(x)
        if n == 0:
            return out
        self.calls += 1
        seed = (int(self.gen.initial_seed()) * 1000003 + self.calls) & 0xFFFFFFFFFFFFFFFF
        self.last_seed = seed                      # seeds the device noise field and the plasma fractals of this call
        lib = _lib.load_library()
        check(lib.jn_augment_patches(ptr(x), ptr(out), ptr(prm), ptr(nz), seed, n, P, _lib.current_stream(x.device)),
              "jn_augment_patches")
        return out
