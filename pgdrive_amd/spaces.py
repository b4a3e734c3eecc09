"""The spaces and the Env base class of the env surface.

With `gym` importable (the reference's dependency: setup.py `gym`, base_env.py:93 `class BasePGDriveEnv(gym.Env)`) the envs of
pgdrive_amd.env / marl_env ARE gym.Env subclasses, their spaces are gym's Box / MultiDiscrete / Dict, and the eight ids of
pgdrive/register.py:5-41 are registered at `import pgdrive_amd` -- `gym.make("PGDrive-v0")` and `isinstance(env, gym.Env)` hold.
Without it (this image has neither gym nor gymnasium) the stand-ins below are used (the reference carries its own copy too:
utils/space.py) and the envs derive from `object`.  gymnasium alone is NOT taken as gym: its Env contract (reset -> (obs, info),
step -> five values) is not the reference's, whose four-value step the envs here return."""
import numpy as np

try:  # the reference's dependency
    import gym as GYM
    import gym.spaces as _gs
except ImportError:  # (stand-ins below)
    GYM = None


class _Box:
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


class _MultiDiscrete:
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


class _Dict(dict):
    """gym.spaces.Dict stand-in for the MARL surface (base_env.py:410-425)."""
    def sample(self):
        return {k: s.sample() for k, s in self.items()}

    def contains(self, x):
        return isinstance(x, dict) and all(k in self and self[k].contains(v) for k, v in x.items())


if GYM is not None:
    Box, MultiDiscrete, Dict, EnvBase = _gs.Box, _gs.MultiDiscrete, _gs.Dict, GYM.Env
else:
    Box, MultiDiscrete, Dict, EnvBase = _Box, _MultiDiscrete, _Dict, object


def register_gym_ids():
    """pgdrive/register.py:5-41: the eight `PGDrive-*-v0` ids -> PGDriveEnv(config=dict(start_seed=..., environment_num=...)).
    Returns the ids registered ([] without gym, or when they already are)."""
    if GYM is None:
        return []
    from gym.envs.registration import register
    from .env import ENV_IDS
    try:
        known = set(getattr(GYM.envs.registry, "env_specs", GYM.envs.registry).keys())  # (a dict in gym >= 0.26, EnvRegistry before)
    except Exception:  # noqa: BLE001
        known = set()
    done = []
    for env_id, env_config in ENV_IDS.items():
        if env_id in known:
            continue
        register(id=env_id, entry_point="pgdrive_amd.env:PGDriveEnv", kwargs=dict(config=dict(env_config)))
        done.append(env_id)
    return done
