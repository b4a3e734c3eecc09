"""The 120-combination run of tests/test_parity_gpu.py::test_marl_config_combinations_fuzz (same generator), continued past differences and\nclassified: flag mismatches per combination, non-ray observation columns off by more than the tolerance (speed cut-off ties named)."""
import sys, os; sys.path.insert(0,'.')
import numpy as np, torch
from tests import util
from tests.test_parity_gpu import OBS_TOL
from oracle import orc
from pgdrive_amd import _abi
from pgdrive_amd.engine import Engine
r = np.random.default_rng(1)
summary = {}; flagm = {}
first = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for trial in range(120):
    kind = str(r.choice(["roundabout", "intersection", "bottleneck", "parking", "pg"]))
    na = int(r.choice([4, 8, 12])); cap = int(r.choice([na, na + 4]))
    if kind == "pg": cap = min(cap, 15); na = min(na, cap)
    if kind == "parking": na = min(na, 10)
    no = int(r.choice([0, 0, 3]))
    kw = dict(crash_done=bool(r.integers(2)), out_of_road_done=bool(r.integers(2)), allow_respawn=bool(r.integers(4) > 0),
              delay_done=int(r.choice([0, 5, 25])), horizon=int(r.choice([60, 150, 1000])), num_others=no,
              others_state=no > 0, side_lasers=int(r.choice([0, 0, 4])), side_dist=50.0,
              lane_line_lasers=int(r.choice([0, 0, 4])), lane_line_dist=20.0, seed=int(r.integers(1000)))
    if kind == "bottleneck": kw.update(plain_reward=True, cross_yellow_line_done=bool(r.integers(2)))
    if kind == "parking": kw.update(parking=True, enable_reverse=True)
    if trial < first: continue
    d, mb, sb = util.make_marl_banks(num_agents=na, capacity=cap, kind=kind)
    n_envs = 24
    cfg = util.marl_config(n_envs, sb, **kw)
    eng, ora = Engine(cfg, mb, sb), orc.Oracle(cfg, mb, sb)
    ids = np.arange(n_envs) % len(sb.scenarios)
    eng.reset(ids); ora.reset(ids)
    rng = np.random.default_rng(trial)
    nl, ks, km = cfg.num_lasers, cfg.side_lasers, cfg.lane_line_lasers
    bad = 0; fm = 0
    for t in range(200):
        act = util.marl_actions(rng, n_envs, sb.A)
        oo, orw, od, ofl = ora.step(act)
        go, grw, gd, gfl = eng.step(torch.from_numpy(act).to(eng.device)); eng.sync()
        go = go.cpu().numpy().astype(np.float64); gfl = gfl.cpu().numpy().astype(np.uint32); gd = gd.cpu().numpy()
        same = (gfl == ofl) & (gd == od)
        fm += int((~same).sum())
        dd = np.abs(go - oo)
        D = dd.shape[2]
        fan = np.zeros(D, bool); fan[:ks] = True; fan[(ks or 2)+6:(ks or 2)+6+km] = True
        if nl: fan[-nl:] = True
        if not (cfg.marl_flags & _abi.MA_OTHERS_STATE):
            x = dd[:, :, ~fan] * same[:, :, None]
            if x.max() > OBS_TOL:
                e, a, c = np.unravel_index(np.argmax(x), x.shape)
                cols = np.nonzero(~fan)[0]
                f, i, ei = ora.get_state()
                speed_tie = cols[c] == (ks or 2) + 1 and abs(abs(f[3, e, a]) * 3.6 - 80.0) < 1.5
                summary.setdefault(trial, []).append("speed cut-off tie" if speed_tie else "OTHER col %d" % cols[c])
                if not speed_tie: print("trial %d %s na %d cap %d step %d env %d agent %d col %d gpu %.6f orc %.6f flags %#x status %d v %.3f kw %s" % (
                    trial, kind, na, cap, t, e, a, cols[c], go[e, a, cols[c]], oo[e, a, cols[c]], ofl[e, a], i[0, e, a], f[3, e, a], kw))
                bad += 1
        f, i, ei = ora.get_state()
        f32 = util.round_state_f32(f); ora.set_state(f32, i, ei); eng.set_state(f32, i, ei)
    eng.close()
    flagm[trial] = fm
print("done", {k: sorted(set(v)) for k, v in summary.items()}, "flag mismatches per trial:", {k: v for k, v in flagm.items() if v})
