"""Teacher-trajectory generator (jolineedle_amd/trajectory.py) against walks recorded from the reference's
NeedleSimpleEnv.generate_sample (tests/golden/g8_trajectories.npz, made by tests/golden/make_golden_trajectories.py),
plus walk invariants on random cases.  Host integer logic: runs without a GPU."""
import random
from pathlib import Path

import numpy as np
import pytest
import torch

from jolineedle_amd.common import ACTION_DELTAS, Action
from jolineedle_amd.trajectory import NeedleSimpleEnv, move_towards

G8 = np.load(Path(__file__).parent / "golden" / "g8_trajectories.npz")
KEYS = ("positions", "current_actions", "next_actions", "labels", "masks", "local_bboxes", "bboxes_yolox")


def case_image(P, gh, gw):
    n = 3 * gh * P * gw * P
    return (torch.arange(n, dtype=torch.float32) / n).reshape(3, gh * P, gw * P)


def run_case(name):
    P, gh, gw, seed, pyseed, T, kmin, kmax, binom = (int(v) for v in G8[f"{name}.args"])
    start = G8[f"{name}.start"]
    env = NeedleSimpleEnv(None, P, G8[f"{name}.boxes"], seed=seed, height=gh * P, width=gw * P)
    random.seed(pyseed)
    s = env.generate_sample_indices(T, kmin, kmax, bool(binom), None if start[0] < 0 else (int(start[0]), int(start[1])))
    return env, s, (P, gh, gw, T)


@pytest.mark.parametrize("name", [str(n) for n in G8["names"]])
def test_walk_matches_reference(name):
    env, s, (P, gh, gw, T) = run_case(name)
    for k in KEYS:
        want = G8[f"{name}.{k}"]
        assert want.shape == s[k].shape and want.dtype == s[k].dtype, (k, want.shape, s[k].shape, want.dtype, s[k].dtype)
        assert np.array_equal(want, s[k]), k
    img = case_image(P, gh, gw).numpy()
    cut = lambda y, x: img[:, y * P:(y + 1) * P, x * P:(x + 1) * P]
    traj = np.stack([cut(y, x) * m for (y, x), m in zip(s["positions"], s["masks"])])
    assert np.array_equal(traj, G8[f"{name}.patches"])
    det = np.stack([cut(y, x) for y, x in s["positions_yolox"]])
    assert np.array_equal(det, G8[f"{name}.patches_yolox"])
    assert np.array_equal(np.array(sorted(env.bbox_patches), np.int64).reshape(-1, 2), G8[f"{name}.bbox_patches"])


def test_move_towards_table():
    for dy in (-3, 0, 2):
        for dx in (-1, 0, 4):
            a = move_towards((5, 5), (5 + dy, 5 + dx))
            assert ACTION_DELTAS[a] == (np.sign(dy), np.sign(dx))
    assert move_towards((2, 2), (2, 2)) is Action.STOP


@pytest.mark.parametrize("seed", range(6))
def test_walk_invariants(seed):
    g = np.random.default_rng(100 + seed)
    P, gh, gw, T = 8, int(g.integers(3, 8)), int(g.integers(3, 8)), 64
    boxes = []
    for _ in range(int(g.integers(1, 4))):
        w, h = int(g.integers(2, 2 * P)), int(g.integers(2, 2 * P))
        x, y = int(g.integers(0, gw * P - w)), int(g.integers(0, gh * P - h))
        boxes.append([x, y, x + w, y + h])
    env = NeedleSimpleEnv(None, P, np.array(boxes), seed=seed, height=gh * P, width=gw * P, py_random=random.Random(seed))
    s = env.generate_sample_indices(T, 0, 3, binomial_keypoints=bool(seed % 2))
    n = int(s["masks"].sum())
    assert 1 <= n <= T and (s["masks"][:n] == 1).all() and (s["masks"][n:] == 0).all()
    pos, cur, nxt = s["positions"], s["current_actions"], s["next_actions"]
    assert (pos[:n, 0] >= 0).all() and (pos[:n, 0] < gh).all() and (pos[:n, 1] >= 0).all() and (pos[:n, 1] < gw).all()
    for t in range(1, n):                                   # each step applies the action taken (clipped to the grid)
        dy, dx = ACTION_DELTAS[Action(int(cur[t]))]
        assert tuple(pos[t]) == (min(max(pos[t - 1][0] + dy, 0), gh - 1), min(max(pos[t - 1][1] + dx, 0), gw - 1))
    assert (nxt[:n] != Action.STOP.value).all()             # the teacher never stops (remove_stop_action)
    visited = {tuple(p) for p in pos[:n]}
    assert env.bbox_patches <= visited                      # every box cell is reached (T is large enough)
    for t in range(n):                                      # labels / local boxes describe the cell of the step
        assert int(s["labels"][t]) == int(tuple(pos[t]) in env.bbox_patches)
        assert np.array_equal(s["local_bboxes"][t], env.local_bboxes(tuple(pos[t])))
    # detector cells: all box cells + exactly one empty cell when one exists
    cells = {tuple(p) for p in s["positions_yolox"]}
    assert env.bbox_patches <= cells and len(cells - env.bbox_patches) == (1 if len(env.bbox_patches) < gh * gw else 0)
    lb = s["bboxes_yolox"]
    assert ((lb[..., 5] == 1) == (lb[..., 3] > lb[..., 1])).all() and (lb[..., 1:5] >= 0).all() and (lb[..., 1:5] <= P).all()


def test_truncation_keeps_last_steps():
    boxes = np.array([[0, 0, 3, 3], [28, 28, 31, 31]])
    a = NeedleSimpleEnv(None, 4, boxes, seed=3, height=32, width=32, py_random=random.Random(1))
    b = NeedleSimpleEnv(None, 4, boxes, seed=3, height=32, width=32, py_random=random.Random(1))
    full = a.generate_sample_indices(64, 0, 0, position=(4, 4))
    short = b.generate_sample_indices(5, 0, 0, position=(4, 4))
    n = int(full["masks"].sum())
    assert n > 5 and short["masks"].sum() == 5
    for k in ("positions", "current_actions", "next_actions", "labels"):
        assert np.array_equal(short[k], full[k][n - 5:n])
