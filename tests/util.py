"""Shared helpers for the parity tests (oracle = checker, HIP engine = product)."""
import numpy as np

from pgdrive_amd import _abi, mapdata, scenario


def make_banks(descs, n_maps=8, num_agents=1, num_traffic=16, density=0.1, first=0):
    sel = descs[first:first + n_maps]
    mb = mapdata.MapBank(sel)
    sb = scenario.ScenarioBank(sel, [d["seed"] for d in sel], num_agents=num_agents, num_traffic=num_traffic,
                               density=density)
    return mb, sb


def driving_actions(rng, n, a=1):
    """Mostly sensible driving (small steering noise, throttle biased forward) so episodes last a while."""
    act = np.zeros((n, a, 2), dtype=np.float32)
    act[..., 0] = np.clip(rng.normal(0, 0.15, size=(n, a)), -1, 1)
    act[..., 1] = np.clip(rng.normal(0.5, 0.5, size=(n, a)), -1, 1)
    return act


def round_state_f32(f):
    return np.asarray(f, dtype=np.float32).astype(np.float64)
