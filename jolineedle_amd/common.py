"""Action vocabulary of the needle environments (mirror of src/env/common.py:4-56).

The (dy, dx) table is also baked into the device step kernel (csrc/jn_device.h);
tests check both against the golden vector g1_env.npz::action_deltas.
"""
from dataclasses import dataclass
from enum import Enum

Action = Enum("Action", ["LEFT", "RIGHT", "UP", "DOWN", "LEFT_UP", "RIGHT_UP",
                         "LEFT_DOWN", "RIGHT_DOWN", "STOP"], start=0)

ACTION_DELTAS = {
    Action.LEFT: (0, -1), Action.RIGHT: (0, 1), Action.UP: (-1, 0), Action.DOWN: (1, 0),
    Action.LEFT_UP: (-1, -1), Action.RIGHT_UP: (-1, 1), Action.LEFT_DOWN: (1, -1),
    Action.RIGHT_DOWN: (1, 1), Action.STOP: (0, 0),
}
MOVES = [a for a in Action if a is not Action.STOP]


@dataclass
class ActionInfo:
    action_type: str
    nclasses: int


def get_actions_info(train_config):
    n = len(Action) if train_config.stop_enabled else len(Action) - 1
    return [ActionInfo(action_type="categorical", nclasses=n)]
