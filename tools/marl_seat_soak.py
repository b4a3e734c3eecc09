"""The seat-folded multi-agent kernels (40 / 44 seats x 72 beams: k_step<..., 40072 | 44072> + k_observe_env<4, ..., 40072 | 44072>; 8 seats x
72 / 240 beams: k_step<..., 8072 | 8240> with the observation fused) against the general
kernels (PGD_NO_FIX=1) over a long run: every step both engines start from the general engine's state and take the same actions
(tests/test_parity_gpu.py::test_default_multi_agent_kernel_matches_the_general_kernel, 300 steps x 32 envs there).  Reports flag / done /
integer-state differences, the largest float differences of the rows that are due, and the corner-grazing beam flips.

    python tools/marl_seat_soak.py [steps=4000] [envs=128]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pgdrive_amd import _abi  # noqa: E402
from pgdrive_amd.engine import Engine  # noqa: E402
from tests import util  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 128
for seats, beams in ((40, 72), (44, 72), (8, 72), (8, 240)):
    d, mb, sb = util.make_marl_banks(num_agents=min(seats, 40), capacity=seats, kind="roundabout")
    cfg = util.marl_config(n_envs, sb, horizon=300, num_lasers=beams)
    os.environ.pop("PGD_NO_FIX", None)
    fix = Engine(cfg, mb, sb)
    os.environ["PGD_NO_FIX"] = "1"
    gen = Engine(cfg, mb, sb)
    os.environ.pop("PGD_NO_FIX", None)
    ids = np.arange(n_envs) % len(sb.scenarios)
    fix.reset(ids); gen.reset(ids)
    rng = np.random.default_rng(17)
    n_flag = n_int = n_graze = n_done = n_new = 0
    rows_due = 0
    worst_obs = worst_rew = worst_state = 0.0
    for t in range(steps):
        act = util.marl_actions(rng, n_envs, sb.A)
        if (t // 500) % 2:  # phases of mostly-throttle actions: the roundabout fills up, agents arrive
            act[..., 1] = np.abs(act[..., 1]); act[..., 0] *= 0.3
        f, i, ei = gen.get_state()
        fix.set_state(f, i, ei)
        a = torch.from_numpy(act).to(gen.device)
        o1, r1, d1, f1 = [x.clone() for x in gen.step(a)]
        o2, r2, d2, f2 = [x.clone() for x in fix.step(a)]
        gen.sync(); fix.sync()
        n_flag += int((f1 != f2).sum()) + int((d1 != d2).sum())
        rep = ((f1 & (_abi.F_REPORT | _abi.F_NEW)) != 0)
        rows_due += int(rep.sum())
        dd = (o1 - o2).abs() * rep[..., None]
        flip = dd[..., 18:] > 1e-4  # a beam through a box corner: hit in one kernel, miss (1.0, or the body behind) in the other
        n_graze += int(flip.sum())
        dd[..., 18:][flip] = 0.0
        worst_obs = max(worst_obs, float(dd.max()))
        worst_rew = max(worst_rew, float(((r1 - r2).abs() * rep).max()))
        g1, i1, e1 = gen.get_state()
        g2, i2, e2 = fix.get_state()
        n_int += int((i1 != i2).sum()) + int((e1 != e2).sum())
        worst_state = max(worst_state, float(np.abs(g1 - g2).max()))
        n_done += int(d1.sum()); n_new += int(((f1 & _abi.F_NEW) != 0).sum())
    print("%d seats x %d beams: %d steps x %d envs, %d rows due, %d finishes, %d (re)spawns | flag/done differences %d, integer-state differences %d, "
          "corner-grazing beam flips %d of %d beams, worst |obs| %.2e, |reward| %.2e, |state| %.2e  [%s]" %
          (seats, beams, steps, n_envs, rows_due, n_done, n_new, n_flag, n_int, n_graze, rows_due * beams, worst_obs, worst_rew, worst_state,
           fix.describe_step()), flush=True)
    fix.close(); gen.close()
