"""Minimal gym-compatible spaces (gym is not a dependency; the reference carries its own copy too: utils/space.py)."""
import numpy as np


class Box:
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low = np.full(shape, low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype)
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    def __contains__(self, x):
        return self.contains(x)

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype)


class MultiDiscrete:
    """gym.spaces.MultiDiscrete stand-in (discrete_action, base_vehicle.py:720-727)."""
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self.dtype = np.dtype(np.int64)
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)

    def sample(self):
        return (self._rng.random_sample(self.nvec.shape) * self.nvec).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= 0)) and bool(np.all(x < self.nvec))

    def __contains__(self, x):
        return self.contains(x)

    def __repr__(self):
        return "MultiDiscrete(%s)" % (self.nvec, )


class Dict(dict):
    """gym.spaces.Dict stand-in for the MARL surface (base_env.py:410-425)."""
    def sample(self):
        return {k: s.sample() for k, s in self.items()}

    def contains(self, x):
        return isinstance(x, dict) and all(k in self and self[k].contains(v) for k, v in x.items())
