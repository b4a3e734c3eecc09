"""Where the throughput-mode kernel (several envs per wave, PGD_PACK=1) and the one-env-per-wave kernel stop agreeing bit for bit: both
engines get the SAME state (set_state from one) before every step, step once, and the state fields / observation columns that differ
are counted by field -- the expression whose fp contraction differs between the two instantiations shows up as the first field of
the step that differs.  usage: mode_diff.py [N] [STEPS]"""
import os
import sys
import numpy as np
sys.path.insert(0, '.')
import torch  # noqa: E402
from pgdrive_amd import _abi, bank, mapdata, scenario  # noqa: E402
from pgdrive_amd.engine import Engine  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 120
descs = bank.get_descriptions(range(1000, 1100))
mb = mapdata.MapBank(descs)
sb = scenario.ScenarioBank(descs, [d["seed"] for d in descs], num_agents=1, num_traffic=16, traffic_mode=os.environ.get("TRAFFIC", "respawn"))
cfg = _abi.make_config(N, auto_reset=1, resample_scenario=1, seed=77)
os.environ["PGD_PACK"] = "1"
pack = Engine(cfg, mb, sb)
os.environ["PGD_PACK"] = "0"
one = Engine(cfg, mb, sb)
ids = (np.arange(N) * 7) % 100
pack.reset(ids); one.reset(ids)
rng = np.random.default_rng(14)
SFn = {v: k for k, v in _abi.SF.items()}
fcount, ocount = {}, {}
worst = {}
for t in range(STEPS):
    a = rng.uniform(-1, 1, size=(N, 1, 2)).astype(np.float32)
    a[:, 0, 1] = np.abs(a[:, 0, 1]) * 0.8
    act = torch.from_numpy(a).cuda()
    f, i, ei = pack.get_state()
    one.set_state(f, i, ei)
    pack.set_state(f, i, ei)  # (both start from the round-tripped state: set_state re-derives what it derives)
    o1 = pack.step(act)[0].clone(); pack.sync()
    o2 = one.step(act)[0].clone(); one.sync()
    if t == 0:
        print(pack.describe_step()); print(one.describe_step())
    f1, i1, e1 = pack.get_state(); f2, i2, e2 = one.get_state()
    same_int = (i1 == i2).all(axis=(0, 2))  # per env
    for k in range(f1.shape[0]):
        d = (f1[k].view(np.uint32) != f2[k].view(np.uint32)) & same_int[:, None]
        if d.any():
            fcount[SFn.get(k, k)] = fcount.get(SFn.get(k, k), 0) + int(d.sum())
            worst[SFn.get(k, k)] = max(worst.get(SFn.get(k, k), 0.0), float(np.abs(f1[k] - f2[k])[d].max()))
    dd = (o1 - o2).abs().view(N, -1).cpu().numpy()[same_int]
    for name, sl in (("state 0..7", slice(0, 8)), ("navi 8..17", slice(8, 18)), ("others 18..33", slice(18, 34)), ("rays 34..", slice(34, None))):
        ocount[name] = ocount.get(name, 0) + int((dd[:, sl] > 0).sum())
        worst["obs " + name] = max(worst.get("obs " + name, 0.0), float(dd[:, sl].max()))
print("state fields whose bits differ after ONE step from identical state (count over %d steps x %d envs x 17 slots):" % (STEPS, N))
for k, v in sorted(fcount.items(), key=lambda kv: -kv[1]):
    print("  %-12s %8d   worst |diff| %.3g" % (k, v, worst[k]))
print("observation columns that differ:", ocount, {k: v for k, v in worst.items() if k.startswith("obs")})
