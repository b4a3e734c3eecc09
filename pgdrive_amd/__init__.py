"""pgdrive_amd — MI355X-native batched PGDrive step engine (HIP kernels behind a C ABI).

Importing the package does not touch the GPU; `PGDriveVecEnv` / `PGDriveEnv` / `Engine` need libpgdrive_hip.so
(built in-tree by `python -m pgdrive_amd.build`) and an MI355X — there is no CPU fallback.
"""
__version__ = "0.1.0"

# pgdrive/__init__.py:1 imports pgdrive.register: the gym ids exist once the package is imported (a no-op without gym)
from .spaces import register_gym_ids as _register_gym_ids  # noqa: E402

_register_gym_ids()


def __getattr__(name):
    if name == "PGDriveVecEnv":
        from .vec_env import PGDriveVecEnv
        return PGDriveVecEnv
    if name in ("PGDriveEnv", "SafePGDriveEnv", "TopDownPGDriveEnv", "TopDownPGDriveEnvV2", "TopDownSingleFramePGDriveEnv", "make"):
        from . import env
        return getattr(env, name)
    if name.startswith("MultiAgent"):  # MultiAgent{Roundabout,Intersection,Bottleneck,Tollgate,ParkingLot}[Vec]Env, MultiAgentPGDrive[VecEnv]
        from . import marl_env
        return getattr(marl_env, name)
    if name == "Engine":
        from .engine import Engine
        return Engine
    raise AttributeError(name)
