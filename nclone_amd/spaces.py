"""Observation/action space descriptions.

Uses gymnasium's spaces when gymnasium is importable (so `isinstance` checks in training code work); otherwise
falls back to tiny stand-ins with the same attribute names (`n`, `shape`, `dtype`, `low`, `high`, `spaces`).
Shapes/dtypes follow the reference: nclone/gym_environment/base_environment.py:150,320-364.
"""
import numpy as np

try:  # pragma: no cover - gymnasium is not installed in the build container
    from gymnasium.spaces import Box, Dict, Discrete  # type: ignore
except Exception:  # noqa: BLE001

    class Discrete:  # type: ignore
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.int64

        def sample(self, rng=None):
            rng = rng or np.random.default_rng()
            return int(rng.integers(0, self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n

        def __repr__(self):
            return "Discrete(%d)" % self.n

    class Box:  # type: ignore
        def __init__(self, low, high, shape, dtype):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

        def __repr__(self):
            return "Box(%r, %r, %r, %s)" % (self.low, self.high, self.shape, self.dtype)

    class Dict:  # type: ignore
        def __init__(self, spaces):
            self.spaces = dict(spaces)

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

        def __repr__(self):
            return "Dict(%r)" % self.spaces


def action_space():
    # 0 NOOP, 1 LEFT, 2 RIGHT, 3 JUMP, 4 JUMP+LEFT, 5 JUMP+RIGHT (base_environment.py:366-402)
    return Discrete(6)


def observation_space(visual=False, spatial_context=False, switch_states=False, reachability=False):
    spaces = {
        "game_state": Box(-1.0, 1.0, (41,), np.float32),
        "action_mask": Box(0, 1, (6,), np.int8),
        "entity_positions": Box(0.0, 1.0, (6,), np.float32),
    }
    if spatial_context:
        spaces["spatial_context"] = Box(-1.0, 1.0, (112,), np.float32)
    if switch_states:
        spaces["switch_states"] = Box(0.0, 1.0, (25,), np.float32)
    if visual:
        spaces["player_frame"] = Box(0, 255, (84, 84, 1), np.uint8)
        spaces["global_view"] = Box(0, 255, (176, 100, 1), np.uint8)   # RENDERED_VIEW_HEIGHT x WIDTH (constants.py:18-19)
    if reachability:   # npp_environment.py observation space: reachability_features (38), mine_sdf_features (3)
        spaces["reachability_features"] = Box(0.0, 1.0, (38,), np.float32)   # the reference declares [0, 1] (npp_environment.py:236)
        spaces["mine_sdf_features"] = Box(-1.0, 1.0, (3,), np.float32)
    return Dict(spaces)
