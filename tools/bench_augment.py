import sys, time, torch
sys.path.insert(0, '.')
from jolineedle_amd.augment import DetectionAugment
dev = "cuda:0"
N, P = 256, 448
x = torch.rand((N, 3, P, P), device=dev)
aug = DetectionAugment(p_planckian=0, seed=1)
prm = aug.sample_params(N)
for _ in range(3): y = aug(x, params=prm)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): y = aug(x, params=prm)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
gb = 2 * x.numel() * 4 / 1e9
print(f"augment {N}x3x{P}x{P}: {ms:.3f} ms  {gb/ms*1e3:.0f} GB/s (algorithmic 8 B/elem)")
