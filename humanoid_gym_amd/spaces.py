"""Minimal stand-in for ``gym.spaces.Box`` used when gym is not installed.

The reference builds ``spaces.Box(low, high)`` for its action and observation
spaces (dancer_env.py:89-90, nao_env.py:45-46, pepper_env.py:53-54) and the
example drivers only call ``action_space.sample()`` (random_control_nao.py:9).
"""
from __future__ import annotations

import numpy as np

try:                                    # real gym when present
    from gym import spaces as _gym_spaces
    Box = _gym_spaces.Box
    HAVE_GYM = True
except Exception:                       # pragma: no cover - depends on the image
    HAVE_GYM = False

    class Box:                          # type: ignore[no-redef]
        def __init__(self, low, high, dtype=np.float32, seed=None):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            assert self.low.shape == self.high.shape
            self.shape, self.dtype = self.low.shape, np.dtype(dtype)
            self._rng = np.random.default_rng(seed)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"
