"""Generate tests/golden/g8_trajectories.npz by running the REFERENCE's NeedleSimpleEnv.generate_sample
(src/env/simple_env.py, imported read-only from /root/reference) on CPU for a set of seeded cases.

Run in the build container only:  python tests/golden/make_golden_trajectories.py
The fixture holds inputs (grid, boxes, seeds, arguments) and the walks the reference produced — no source.
Images are the deterministic ramp `case_image()` below so the tests can rebuild them.
"""
import random
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.dont_write_bytecode = True

from make_golden import install_stubs, REF  # noqa: E402

# (name, patch, grid_h, grid_w, boxes xyxy, np seed, py seed, max_ep_len, min_kp, max_kp, binomial, start | None)
CASES = [
    ("one_box", 4, 5, 6, [[9, 5, 14, 11]], 0, 0, 12, 0, 0, False, (0, 0)),
    ("two_boxes_random_start", 4, 6, 7, [[1, 1, 7, 3], [18, 14, 27, 23]], 1, 5, 24, 0, 2, False, None),
    ("ties", 4, 7, 7, [[0, 12, 3, 15], [24, 12, 27, 15], [12, 0, 15, 3], [12, 24, 15, 27]], 2, 11, 40, 0, 0, False, (3, 3)),
    ("detours", 4, 6, 6, [[4, 4, 11, 11]], 3, 7, 30, 2, 4, False, (5, 5)),
    ("binomial", 4, 8, 8, [[13, 2, 18, 9], [25, 25, 31, 31]], 4, 9, 40, 1, 3, True, None),
    ("truncated", 4, 8, 8, [[0, 0, 3, 3], [28, 28, 31, 31], [0, 28, 3, 31]], 5, 13, 6, 0, 1, False, (4, 4)),
    ("start_inside", 8, 4, 5, [[10, 9, 30, 20]], 6, 17, 16, 0, 1, False, (1, 2)),
    ("no_boxes", 4, 4, 4, [], 7, 19, 8, 0, 0, False, None),
    ("box_partly_outside", 4, 4, 4, [[12, 12, 19, 18]], 8, 23, 10, 0, 1, False, (0, 0)),
    ("thin_box_below_area_threshold", 10, 4, 4, [[9, 9, 21, 10]], 9, 29, 10, 0, 0, False, (3, 3)),
]


def case_image(P, gh, gw):
    n = 3 * gh * P * gw * P
    return (torch.arange(n, dtype=torch.float32) / n).reshape(3, gh * P, gw * P)


def main():
    install_stubs()
    sys.path.insert(0, str(REF))
    from src.env.simple_env import NeedleSimpleEnv
    from src.utils import BBox, Position

    out = {"names": np.array([c[0] for c in CASES])}
    for name, P, gh, gw, boxes, seed, pyseed, T, kmin, kmax, binom, start in CASES:
        image = case_image(P, gh, gw)
        bbs = [BBox(up_left=Position(y=b[1], x=b[0]), bottom_right=Position(y=b[3], x=b[2])) for b in boxes]
        env = NeedleSimpleEnv(image, P, bbs, seed=seed)
        random.seed(pyseed)
        pos = None if start is None else Position(*start)
        s = env.generate_sample(T, kmin, kmax, binomial_keypoints=binom, position=pos)
        out[f"{name}.args"] = np.array([P, gh, gw, seed, pyseed, T, kmin, kmax, int(binom)], np.int64)
        out[f"{name}.boxes"] = np.array(boxes, np.int64).reshape(-1, 4)
        out[f"{name}.start"] = np.array(start if start is not None else (-1, -1), np.int64)
        for k, v in s.items():
            out[f"{name}.{k}"] = v.numpy()
        # deterministic helpers on the same env
        out[f"{name}.bbox_patches"] = np.array(sorted(env.bbox_patches), np.int64).reshape(-1, 2)
    np.savez_compressed(HERE / "g8_trajectories.npz", **out)
    print("g8_trajectories.npz", (HERE / "g8_trajectories.npz").stat().st_size)


if __name__ == "__main__":
    main()
