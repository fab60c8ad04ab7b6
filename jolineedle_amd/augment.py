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
  the CIED table itself is data of kornia — pass it as ``planckian_coeffs`` [K, 2]; without it the op is skipped.
* PlasmaShadow needs kornia's diamond-square fractal generator and is not implemented.
"""
import math
from typing import Optional

import torch

from . import _lib
from ._lib import check, ptr

NPARAM = 16


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
                 p_motion: float = 0.3, angle=(-180.0, 180.0), seed: Optional[int] = None):
        self.planckian = planckian_coeffs
        self.p = (p_planckian, p_gray, p_blur, p_noise, p_motion)
        self.sigma, self.noise_std, self.angle = sigma, noise_std, angle
        self.gen = torch.Generator()
        if seed is not None:
            self.gen.manual_seed(seed)
        self.calls = 0

    def sample_params(self, n: int) -> torch.Tensor:
        """[n, NPARAM] = r_gain, b_gain, gray, w0, w1, noise_std, k[9], pad — identity where an op was not drawn."""
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
        lib = _lib.load_library()
        check(lib.jn_augment_patches(ptr(x), ptr(out), ptr(prm), ptr(nz), seed, n, P, _lib.current_stream(x.device)),
              "jn_augment_patches")
        return out
