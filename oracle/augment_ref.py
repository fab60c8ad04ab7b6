"""CPU restatement (test infrastructure only) of the detection-augmentation arithmetic of
jolineedle_amd/augment.py / csrc/kernels_aug.hip: the op chain of Trainer.init_detection (src/trainer.py:176-186)
for GIVEN per-patch parameters — colour gains + clamp, grayscale (0.299, 0.587, 0.114), 3x3 Gaussian with reflect
padding, additive noise, 3x3 motion kernel with zero padding — written with torch's own conv / pad ops.
kornia (the reference's provider of these ops) is not in the reference tree: parity unpinned."""
import torch
import torch.nn.functional as F


def augment_ref(x: torch.Tensor, params: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
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
        img = img + p[5] * noise[i]
        km = p[6:15].reshape(1, 1, 3, 3).repeat(3, 1, 1, 1)
        out[i] = F.conv2d(img[None], km, padding=1, groups=3)[0]
    return out
