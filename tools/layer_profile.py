#!/usr/bin/env python3
"""Per-layer timing table of one PAFPN pass (64 patches of 448 px): run with JN_LAYER_PROFILE=1, the library prints
one line per op (HIP events around every launch) on stderr.   usage: JN_LAYER_PROFILE=1 python tools/layer_profile.py [f32|bf16] [train]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import jolineedle_amd as ja
from jolineedle_amd.config import model_config
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
train = len(sys.argv) > 2 and sys.argv[2] == "train"
P, N = 448, 64
m = ja.GPT(model_config(patch_size=P, block_size=20, with_detector=False, image_processor=None, act_dtype=dt), max_batch=N)
x = torch.rand(N, 3, P, P, device="cuda")

for i in range(3):
    m.backbone_features(x, train=train)
torch.cuda.synchronize()
