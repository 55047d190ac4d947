"""Minimal stand-ins for the two gym spaces the hot path touches (gym is not a dependency of this package).

The reference reads only ``len(space)``, ``space[i]``, ``.shape``, ``.low/.high`` and ``.dtype``
(runner.py:14-16, policies.py:48,142, agents.py:85-115).
"""
import numpy as np


class Box:
    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def sample(self, rng=None):
        rng = rng or np.random
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return rng.uniform(lo, hi).astype(self.dtype)

    def __repr__(self):
        return "Box%s" % (self.shape,)


class Tuple:
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __iter__(self):
        return iter(self.spaces)

    def sample(self, rng=None):
        return tuple(s.sample(rng) for s in self.spaces)

    def __repr__(self):
        return "Tuple(%s)" % ", ".join(map(repr, self.spaces))
