"""Shared helpers for the parity tests (oracle = checker, HIP engine = product)."""
import numpy as np

from pgdrive_amd import _abi, mapdata, scenario


def make_banks(descs, n_maps=8, num_agents=1, num_traffic=16, density=0.1, first=0, **kw):
    sel = descs[first:first + n_maps]
    mb = mapdata.MapBank(sel)
    sb = scenario.ScenarioBank(sel, [d["seed"] for d in sel], num_agents=num_agents, num_traffic=num_traffic,
                               density=density, **kw)
    return mb, sb


def driving_actions(rng, n, a=1):
    """Mostly sensible driving (small steering noise, throttle biased forward) so episodes last a while."""
    act = np.zeros((n, a, 2), dtype=np.float32)
    act[..., 0] = np.clip(rng.normal(0, 0.15, size=(n, a)), -1, 1)
    act[..., 1] = np.clip(rng.normal(0.5, 0.5, size=(n, a)), -1, 1)
    return act


def round_state_f32(f):
    return np.asarray(f, dtype=np.float32).astype(np.float64)


def make_marl_banks(num_agents=8, n_variants=8, seed=1, capacity=None, kind="roundabout"):
    from pgdrive_amd import mapgen
    if kind == "pg":  # the generic MultiAgentPGDrive over generated maps (multi_agent_pgdrive.py:12-55)
        from pgdrive_amd import bank
        d = bank.get_descriptions([3, 4, 5, 6])
        mb = mapdata.MapBank(d, truncate_succ=True)
        sb = scenario.MarlScenarioBank(d, num_agents=num_agents, capacity=capacity, n_variants=max(1, n_variants // 4), seed=seed,
                                       kind=kind)
        return d, mb, sb
    d = dict(roundabout=mapgen.generate_ma_roundabout, intersection=mapgen.generate_ma_intersection,
             bottleneck=mapgen.generate_ma_bottleneck, tollgate=mapgen.generate_ma_tollgate,
             parking=mapgen.generate_ma_parking_lot)[kind]()
    mb = mapdata.MapBank([d], truncate_succ=True)  # no IDM traffic on the multi-agent maps
    sb = scenario.MarlScenarioBank(d, num_agents=num_agents, capacity=capacity, n_variants=n_variants, seed=seed, kind=kind)
    return d, mb, sb


def marl_config(n_envs, sb, **kw):
    """MULTI_AGENT_PGDRIVE_DEFAULT_CONFIG (multi_agent_pgdrive.py:12-55): 72 beams / 40 m / 0 others, penalties 10."""
    args = dict(num_agents=sb.A, num_traffic=sb.B, num_lasers=72, num_others=0, lidar_dist=40.0, multi_agent=True, horizon=1000,
                agent_limit=sb.num_agents, respawn_places=sb.P, respawn_dests=sb.Dn, out_of_road_penalty=10.0,
                crash_vehicle_penalty=10.0, crash_object_penalty=10.0, delay_done=25, auto_reset=1)
    args.update(kw)
    return _abi.make_config(n_envs, **args)


def marl_actions(rng, n, a):
    act = np.zeros((n, a, 2), dtype=np.float32)
    act[..., 0] = np.clip(rng.normal(0, 0.25, size=(n, a)), -1, 1)
    act[..., 1] = np.clip(rng.normal(0.6, 0.4, size=(n, a)), -1, 1)
    return act


# ---------------------------------------------------------------------------------------------------------------------
# every float field of the state (include/pgd_state_layout.h), GPU (fp32) vs oracle (fp64) after ONE teacher-forced step from
# an identical fp32 state: |gpu - oracle| <= atol + rtol * |oracle|
# ---------------------------------------------------------------------------------------------------------------------
STATE_TOL = {
    "X": (1e-3, 0.0), "Y": (1e-3, 0.0),            # lane coordinates <= 1e-3 m on maps of <= 500 m extent
    "THETA": (1e-4, 0.0), "SPEED": (1e-4, 0.0),
    # the applied action: agents copy the clipped input (exact); traffic stores the PID / IDM-law output, whose inputs are
    # differences of ~100 m lane coordinates in fp32 (steering up to ~10, acceleration down to ~ -100)
    "STEER": (1e-4, 2e-4), "ACT1S": (1e-4, 2e-4),
    # the IDM law divides by the gap to the leader (floored at 1 cm, idm_policy.py:254-271): behind a vehicle a few cm ahead the
    # acceleration is -(d* / gap)^2 ~ -1e5 and twice as ill-conditioned as the gap, a difference of two fp32 lane coordinates
    "THROTTLE": (1e-4, 1e-3), "ACT1T": (1e-4, 1e-3),
    "ACT0S": (1e-6, 0.0), "ACT0T": (1e-6, 0.0),    # the older deque entry is a copy of the previous state's newer one
    "LASTX": (1e-6, 0.0), "LASTY": (1e-6, 0.0),    # copies of the pose the step started from
    "LASTHX": (1e-6, 0.0), "LASTHY": (1e-6, 0.0),
    "PID_HP": (2e-5, 0.0), "PID_LP": (2e-5, 0.0),  # the errors themselves: a heading difference [rad], a lateral offset [m]
    "PID_HI": (2e-5, 1e-4), "PID_LI": (2e-5, 1e-4),  # their running sums
    "TARGET_SPEED": (0.0, 0.0),                    # 30 / 5 km/h
    # a step's energy is proportional to the displacement, a difference of two ~100 m coordinates in fp32 (1 ulp = 1.5e-5 m
    # on a ~2 m step at 80 km/h = 1e-5 relative), and early in an episode the sum IS the last step
    "ENERGY": (2e-6, 2e-5),
    "DIST_LEFT": (1e-4, 0.0), "DIST_RIGHT": (1e-4, 0.0),
    "EP_REWARD": (2e-4, 1e-5),
    "AGENT_ID": (0.0, 0.0),
    "HX": (5e-6, 0.0), "HY": (5e-6, 0.0),          # carried unit heading vector vs cos / sin of the oracle's angle
}


def idm_tie(gf, f):
    """Slots whose applied throttle differs between engine and oracle by more than rounding: an IDM leader exactly MAX_DIST =
    30 m ahead on the 10 m spawn grid is found / not found by the last bit of a lane coordinate (also in the reference's
    fp64), and the vehicle then gets another acceleration on the two sides.  The enumerated tie class of the campaigns
    (profiles/r01_parity_campaign.md): counted and bounded by the callers, excluded from the per-field comparison."""
    a, b = gf[_abi.SF["ACT1T"]].astype(np.float64), np.asarray(f[_abi.SF["ACT1T"]], dtype=np.float64)
    return np.abs(a - b) > 1e-3 + 1e-3 * np.abs(b)


def compare_state(gf, f, mask, worst, skip=()):
    """All PGD_NF float fields of `gf` (engine) against `f` (oracle) on the slots selected by `mask` [N, V].  `worst` maps
    field -> largest error seen so far in units of its tolerance (<= 1 passes); returns it."""
    if not mask.any():
        return worst
    for name, k in _abi.SF.items():
        if name in skip:
            continue
        atol, rtol = STATE_TOL[name]
        a, b = gf[k].astype(np.float64)[mask], np.asarray(f[k], dtype=np.float64)[mask]
        d = np.abs(a - b)
        if name == "THETA":  # heading_theta lives in [-3 pi / 2, pi / 2): a value on the seam may wrap on one side only
            d = np.minimum(d, np.abs(d - 2 * np.pi))
        tol = atol + rtol * np.abs(b)
        if atol == 0.0 and rtol == 0.0:
            err = float((d != 0).any()) * 2.0
        else:
            err = float((d / tol).max())
        worst[name] = max(worst.get(name, 0.0), err)
    return worst


def state_failures(worst):
    return {k: round(v, 3) for k, v in worst.items() if v > 1.0}


def write_fake_gym(root):
    """A minimal `gym` package (Env, spaces.Box / MultiDiscrete / Dict, envs.registration.register / make / registry) under `root`:
    this image has no gym, and the env surface's gym identity (pgdrive_amd/spaces.py) is decided at import -- the tests run a
    subprocess with `root` on PYTHONPATH.  Not a stand-in for gym's behaviour: just enough surface to see what the package does
    with a gym it finds."""
    import os
    g = os.path.join(str(root), "gym")
    os.makedirs(os.path.join(g, "envs"), exist_ok=True)
    open(os.path.join(g, "__init__.py"), "w").write(
        "class Env:\n    metadata = {}\n    def reset(self):\n        raise NotImplementedError\n"
        "    def step(self, action):\n        raise NotImplementedError\n"
        "from . import spaces, envs\nfrom .envs.registration import make, register\n__version__ = '0.0-fake'\n")
    open(os.path.join(g, "spaces.py"), "w").write(
        "import numpy as np\n"
        "class Space:\n    pass\n"
        "class Box(Space):\n    def __init__(self, low, high, shape=None, dtype=np.float32):\n"
        "        self.low = np.full(shape, low, dtype=dtype); self.high = np.full(shape, high, dtype=dtype)\n"
        "        self.shape = tuple(shape); self.dtype = np.dtype(dtype)\n"
        "    def contains(self, x):\n        x = np.asarray(x); return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())\n"
        "class MultiDiscrete(Space):\n    def __init__(self, nvec):\n        self.nvec = np.asarray(nvec); self.shape = self.nvec.shape\n"
        "class Dict(Space):\n    def __init__(self, spaces):\n        self.spaces = dict(spaces)\n"
        "    def keys(self):\n        return self.spaces.keys()\n    def __getitem__(self, k):\n        return self.spaces[k]\n")
    open(os.path.join(g, "envs", "__init__.py"), "w").write("from .registration import registry, register, make\n")
    open(os.path.join(g, "envs", "registration.py"), "w").write(
        "import importlib\nregistry = {}\n"
        "def register(id, entry_point=None, kwargs=None, **more):\n    registry[id] = dict(entry_point=entry_point, kwargs=dict(kwargs or {}))\n"
        "def make(id, **kw):\n    spec = registry[id]; ep = spec['entry_point']\n"
        "    if isinstance(ep, str):\n        mod, attr = ep.split(':'); ep = getattr(importlib.import_module(mod), attr)\n"
        "    return ep(**dict(spec['kwargs'], **kw))\n")
    return str(root)
