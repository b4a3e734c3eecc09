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
