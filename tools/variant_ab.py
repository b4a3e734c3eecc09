"""Step time of the env classes' configurations with their specialised k_step instantiation against the general kernel (PGD_NO_FIX=1):
the top-down envs (lidar off; with and without the state row) and SafePGDriveEnv (16 traffic + 40 object slots).
usage: variant_ab.py [N] [STEPS]"""
import os
import sys

import numpy as np

sys.path.insert(0, '.')
import torch  # noqa: E402
from pgdrive_amd import _abi, bank, mapdata, scenario  # noqa: E402
from pgdrive_amd.engine import Engine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
descs = bank.get_descriptions(range(1000, 1100))
mb = mapdata.MapBank(descs)
rng = np.random.default_rng(0)
acts = torch.from_numpy(rng.uniform(-1, 1, size=(64, N, 1, 2)).astype(np.float32)).cuda()
seeds = [d["seed"] for d in descs]
VARIANTS = {
    "default": (dict(num_traffic=16), dict(num_lasers=240), True),
    "top_down (state row written)": (dict(num_traffic=16), dict(num_lasers=0), True),
    "top_down (no row: bench --topdown)": (dict(num_traffic=16), dict(num_lasers=0), False),
    "safe (accident_prob 0.8, 56 slots)": (dict(num_traffic=56, density=0.05, accident_prob=0.8), dict(num_lasers=240, safe_rl_env=True, use_lateral=False), True),
}
VARIANTS["72 beams, 12 traffic slots, 2 neighbours (no instantiation in the library)"] = (
    dict(num_traffic=12), dict(num_lasers=72, num_others=2), True)
VARIANTS["default + discrete actions + lidar noise (general row layout)"] = (
    dict(num_traffic=16), dict(num_lasers=240, discrete_action=True, lidar_gaussian_noise=0.02), True)
for name, (skw, ckw, want_obs) in VARIANTS.items():
    sb = scenario.ScenarioBank(descs, seeds, num_agents=1, **skw)
    for no_fix in (False, True, "jit"):
        if no_fix is True:
            os.environ["PGD_NO_FIX"] = "1"
        else:
            os.environ.pop("PGD_NO_FIX", None)
        cfg = _abi.make_config(N, num_agents=1, num_traffic=skw["num_traffic"], auto_reset=1, seed=1234, **ckw)
        eng = Engine(cfg, mb, sb)
        if no_fix == "jit":  # a step kernel built for this engine at run time (pgdrive_amd/jit.py); only where the library runs a general kernel
            if not eng.specialise(wait=True):
                eng.close()
                continue
        eng.reset(np.arange(N) % 100)
        with torch.cuda.stream(eng.stream):
            for k in range(1500):
                eng.step(acts[k % 64], want_obs=want_obs)
            eng.sync()
            eng.profile_begin(STEPS // 64 + 2, stride=64)
            for k in range(STEPS):
                eng.step(acts[k % 64], want_obs=want_obs)
            eng.sync()
            p = eng.profile_end()
        print("%-38s %-8s k_step %.2f us   %s" % (name[:38], {False: "shipped", True: "general", "jit": "run-time"}[no_fix], p["k_step_ms"] * 1e3,
                                                  eng.describe_step()), flush=True)
        eng.close()
os.environ.pop("PGD_NO_FIX", None)
