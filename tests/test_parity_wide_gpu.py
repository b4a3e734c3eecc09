"""Wider GPU parity runs (through the C ABI, oracle = checker): all 100 PGDrive-v0 maps with three action streams, BASELINE
configurations C2 (1024 envs, several envs per wave) and C5 (multi-agent roundabout with 240 beams, D = 258; 4096 x 8 at
full size through properties), and an IEEE-arithmetic build of the engine as A/B for the discrete outcomes."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from pgdrive_amd import _abi
from tests import util
from tests.test_parity_gpu import OBS_TOL, REW_TOL, STATE_OBS_TOL, _compare_step

pytestmark = pytest.mark.gpu
THREADS = min(64, len(os.sched_getaffinity(0)))


def _actions(mode, rng, n):
    if mode == "driving":
        return util.driving_actions(rng, n)
    if mode == "uniform":
        return rng.uniform(-1, 1, size=(n, 1, 2)).astype(np.float32)
    act = np.zeros((n, 1, 2), np.float32)  # drive straight, full throttle
    act[..., 1] = 1.0
    act[..., 0] = rng.normal(0, 0.05, size=(n, 1))
    return act


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("mode,steps", [("driving", 260), ("uniform", 120), ("straight", 200)])
def test_all_maps_campaign(mode, steps):
    """Teacher-forced GPU vs oracle on ALL 100 PGDrive-v0 maps (8 envs per map), 17 slots, 240 beams, auto-reset onto a
    re-drawn scenario: done / flags
    and the integer state bit-exact; observations, rewards and poses within tolerance.  Discrete fp32-vs-fp64 ties are
    counted by class, as in the one-off campaign of round 1 (profiles/r01_parity_campaign.md, 3.07 M env-steps, 0 flag
    mismatches): grazing lidar beams, a body exactly on the 50 m neighbour radius, an IDM leader exactly MAX_DIST = 30 m
    ahead on the 10 m spawn grid."""
    import torch
    from oracle import orc
    from pgdrive_amd import bank
    from pgdrive_amd.engine import Engine
    descs = bank.get_descriptions(range(1000, 1100))
    n_envs = 800
    mb, sb = util.make_banks(descs, n_maps=100)
    cfg = _abi.make_config(n_envs, num_agents=1, num_traffic=16, num_lasers=240, auto_reset=1, seed=11, resample_scenario=1)
    eng = Engine(cfg, mb, sb)
    ora = orc.Oracle(cfg, mb, sb)
    ids = np.arange(n_envs) % 100
    o0 = ora.reset(ids)
    g0 = eng.reset(ids).cpu().numpy()
    assert np.abs(g0 - o0).max() < OBS_TOL
    rng = np.random.default_rng(17)
    st = dict(steps=0, flag_mismatch=0, obs=0.0, rew=0.0, pose=0.0, beams=0, grazing=0, int_mismatch=0, done=0, active=0,
              radius_rows=0, idm_ties=0)
    worst = {}
    for t in range(steps):
        act = _actions(mode, rng, n_envs)
        oo, orw, od, ofl = ora.step(act, threads=THREADS)
        go, grw, gd, gfl = eng.step(torch.from_numpy(act).to(eng.device))
        eng.sync()
        go = go.cpu().numpy().astype(np.float64)
        grw = grw.cpu().numpy().astype(np.float64)
        gd, gfl = gd.cpu().numpy(), gfl.cpu().numpy().astype(np.uint32)
        same = (gfl == ofl) & (gd == od)
        st["steps"] += same.size
        st["flag_mismatch"] += int((~same).sum())
        st["done"] += int(od.sum())
        d = np.abs(go - oo)[same]
        nb = d[:, 34:]
        graze = nb > OBS_TOL
        st["beams"] += nb.size
        st["grazing"] += int(graze.sum())
        head = d[:, :34]
        flip = (head[:, 18:].max(axis=1) > OBS_TOL) & (head[:, :18].max(axis=1) <= OBS_TOL)  # neighbour block alone differs
        st["radius_rows"] += int(flip.sum())
        st["obs"] = max(st["obs"], float(head[~flip].max()), float(nb[~graze].max()))
        st["obs_state"] = max(st.get("obs_state", 0.0), float(head[~flip].max()))  # state + navigation + neighbour columns alone
        st["rew"] = max(st["rew"], float(np.abs(grw - orw)[same].max()))
        f, i, ei = ora.get_state()
        gf, gi, gei = eng.get_state()
        agree = (gi == i).all(axis=0) & (gei == ei).all(axis=0)[:, None]
        st["int_mismatch"] += int((~agree).sum())
        st["active"] += int((i[0, :, 1:] == 2).sum())
        tie = util.idm_tie(gf, f)
        st["idm_ties"] += int((tie & agree).sum())
        for fld in ("X", "Y", "THETA", "SPEED"):
            dd = np.abs(gf[_abi.SF[fld]].astype(np.float64) - f[_abi.SF[fld]])[agree & ~tie]
            if fld == "THETA":
                dd = np.minimum(dd, np.abs(dd - 2 * np.pi))
            st["pose"] = max(st["pose"], float(dd.max()))
        util.compare_state(gf, f, agree & ~tie, worst)  # all 26 float fields, not the pose alone
        f32 = util.round_state_f32(f)
        ora.set_state(f32, i, ei)
        eng.set_state(f32, i, ei)
    print("campaign", mode, st, "state fields (x tolerance):", {k: round(v, 3) for k, v in worst.items()})
    eng.close()
    assert not util.state_failures(worst), util.state_failures(worst)
    assert st["flag_mismatch"] == 0  # bit-exact done / collision / line / sidewalk / arrive flags
    assert st["int_mismatch"] <= 1   # lane picks on a box edge (1 in 3.07 M in the round-1 campaign)
    assert st["obs"] < OBS_TOL and st["obs_state"] < STATE_OBS_TOL and st["rew"] < REW_TOL and st["pose"] < 1e-3
    assert st["grazing"] <= 1e-6 * st["beams"] + 3 and st["radius_rows"] <= 2
    assert st["idm_ties"] <= 2e-3 * max(st["active"], 1) + 2
    if mode != "uniform":
        assert st["done"] > 200 and st["active"] > 50000


def test_c2_1024_envs_parity():
    """BASELINE C2: 1024 envs, 1 ego, no traffic, no lidar (D = 18) on all 100 maps -- the engine then packs several envs
    into a wave (k_step<ONE_ENV = false>) and runs the stand-alone observation kernel."""
    import torch
    from oracle import orc
    from pgdrive_amd import bank
    from pgdrive_amd.engine import Engine
    descs = bank.get_descriptions(range(1000, 1100))
    n_envs = 1024
    mb, sb = util.make_banks(descs, n_maps=100, num_traffic=0)
    cfg = _abi.make_config(n_envs, num_agents=1, num_traffic=0, num_lasers=0, auto_reset=1, seed=3, resample_scenario=1)
    eng = Engine(cfg, mb, sb)
    ora = orc.Oracle(cfg, mb, sb)
    assert eng.D == 18
    ids = np.arange(n_envs) % 100
    assert np.abs(eng.reset(ids).cpu().numpy() - ora.reset(ids)).max() < OBS_TOL
    rng = np.random.default_rng(2)
    stats = dict(steps=0, flag_mismatch=0, obs=0.0, rew=0.0)
    pose = 0.0
    n_done = 0
    worst = {}
    for t in range(220):
        act = util.driving_actions(rng, n_envs) if t % 2 else _actions("straight", rng, n_envs)
        n_done += int(_compare_step(torch, eng, ora, act, stats).sum())
        f, i, ei = ora.get_state()
        gf, gi, gei = eng.get_state()
        assert (gi == i).all() and (gei == ei).all()
        for fld in ("X", "Y", "SPEED"):
            pose = max(pose, float(np.abs(gf[_abi.SF[fld]].astype(np.float64) - f[_abi.SF[fld]]).max()))
        util.compare_state(gf, f, np.ones(gi.shape[1:], dtype=bool), worst)
        f32 = util.round_state_f32(f)
        ora.set_state(f32, i, ei)
        eng.set_state(f32, i, ei)
    print("C2 parity:", stats, "pose", pose, "episodes", n_done, "state fields (x tolerance):", {k: round(v, 3) for k, v in worst.items()})
    eng.close()
    assert not util.state_failures(worst), util.state_failures(worst)
    assert stats["flag_mismatch"] == 0 and stats["obs"] < STATE_OBS_TOL and stats["rew"] < REW_TOL and pose < 1e-3 and n_done > 300


def test_c5_marl_240_beams_parity():
    """BASELINE C5's row: 8 agents on the multi-agent roundabout with the single-agent lidar (240 beams x 50 m, D = 258)."""
    from tests.test_parity_gpu import test_marl_roundabout_parity
    test_marl_roundabout_parity(8, 8, num_lasers=240, lidar_dist=50.0)


@pytest.mark.timeout(900)
def test_c5_full_size_properties():
    """BASELINE C5 at full size (4096 envs x 8 agents, 72 beams) through size-independent properties: rows inside [0, 1];
    the first 16 envs of the batch equal a 16-env engine bit for bit (envs do not interact; RNG streams keyed per env); the
    run is reproducible; REPORT rows are exactly the agents that were active, NEW rows carry reward 0 and are not done;
    ALL_DONE comes with RESET; agent ids grow monotonically per env."""
    import torch
    from pgdrive_amd.engine import Engine
    d, mb, sb = util.make_marl_banks(num_agents=8, n_variants=16, seed=2)
    N, n, A = 4096, 16, 8

    def make(n_envs):
        return Engine(util.marl_config(n_envs, sb, horizon=150, resample_scenario=1, seed=77), mb, sb)

    big, small, twin = make(N), make(n), make(N)
    ids = (np.arange(N) * 5) % len(sb.scenarios)
    ob = big.reset(ids).clone()
    assert torch.equal(ob[:n], small.reset(ids[:n]))
    twin.reset(ids)
    rng = np.random.default_rng(9)
    seen = dict(report=0, new=0, all_done=0, done=0)
    last_id = np.full(N, -1.0)
    for t in range(200):
        act = torch.from_numpy(util.marl_actions(rng, N, A)).to(big.device)
        f_before = big.get_state()[1][_abi.SI["STATUS"]] if t % 50 == 0 else None
        o1, r1, d1, f1 = [x.clone() for x in big.step(act)]
        o2, r2, d2, f2 = small.step(act[:n].contiguous())
        o3, r3, d3, f3 = twin.step(act)
        big.sync(); small.sync(); twin.sync()
        assert torch.isfinite(o1).all() and float(o1.min()) >= 0.0 and float(o1.max()) <= 1.0
        assert torch.equal(o1[:n], o2) and torch.equal(r1[:n], r2) and torch.equal(d1[:n], d2) and torch.equal(f1[:n], f2)
        assert torch.equal(o1, o3) and torch.equal(r1, r3) and torch.equal(f1, f3)
        fl = f1.cpu().numpy().astype(np.uint32)
        rw, dn = r1.cpu().numpy(), d1.cpu().numpy()
        rep, new, alld, rst = [(fl & b) != 0 for b in (_abi.F_REPORT, _abi.F_NEW, _abi.F_ALL_DONE, _abi.F_RESET)]
        if f_before is not None:
            assert (rep == (f_before[:, :A] == _abi.ST_ACTIVE)).all()
        assert (rw[new & ~rep] == 0.0).all() and (dn[new & ~rep] == 0).all() and (dn[~rep] == 0).all()
        assert (alld.any(axis=1) == rst.any(axis=1)).all() and (alld.all(axis=1) == alld.any(axis=1)).all()
        seen["report"] += int(rep.sum()); seen["new"] += int(new.sum()); seen["all_done"] += int(alld[:, 0].sum())
        seen["done"] += int(dn.sum())
        if t % 25 == 0:
            ff, ii, ee = big.get_state()
            nxt = ee[_abi.EI["NEXT_AGENT"]].astype(np.float64)
            act_ids = np.where(ii[_abi.SI["STATUS"]][:, :A] == _abi.ST_ACTIVE, ff[_abi.SF["AGENT_ID"]][:, :A], -1.0)
            assert (act_ids.max(axis=1) < nxt).all()
    print("C5 full size:", seen)
    assert seen["report"] > 3_000_000 and seen["new"] > 10000 and seen["all_done"] > 1000 and seen["done"] > 10000
    for e in (big, small, twin):
        e.close()


@pytest.mark.timeout(900)
def test_ieee_build_gives_the_same_discrete_outcomes(descs):
    """The shipped library is built with fast fp32 division / sqrt, reciprocal math and denormal flush (pgdrive_amd/build.py):
    every flag is a threshold on such arithmetic.  A/B against a build with IEEE division / sqrt and no fast-math flag on the
    same teacher-forced inputs: done / flags / integer state of the two builds are identical, and both agree with the
    fp64 oracle."""
    import torch
    from oracle import orc
    from pgdrive_amd import build, engine
    td = tempfile.mkdtemp(prefix="pgd_ieee_")
    lib = os.path.join(td, "libpgdrive_hip_ieee.so")
    subprocess.check_call([build.hipcc(), "--offload-arch=gfx950", *build.OPT, "-std=c++17", "-shared", "-fPIC", "-o", lib, build.SRC])
    L_ieee = engine.load_library(path=lib)
    n_envs = 256
    mb, sb = util.make_banks(descs, n_maps=8)
    cfg = _abi.make_config(n_envs, num_agents=1, num_traffic=16, num_lasers=240, auto_reset=1, seed=4)
    fast = engine.Engine(cfg, mb, sb)
    ieee = engine.Engine(cfg, mb, sb, lib=L_ieee)
    ora = orc.Oracle(cfg, mb, sb)
    ids = np.arange(n_envs) % 8
    ora.reset(ids); fast.reset(ids); ieee.reset(ids)
    rng = np.random.default_rng(8)
    diff_flags = diff_ints = vs_oracle = n_done = 0
    worst = 0.0
    for t in range(300):
        act = util.driving_actions(rng, n_envs) if t % 3 else _actions("straight", rng, n_envs)
        oo, orw, od, ofl = ora.step(act, threads=THREADS)
        a = torch.from_numpy(act).to(fast.device)
        fo, frw, fd, ffl = [x.clone() for x in fast.step(a)]
        io, irw, idn, ifl = [x.clone() for x in ieee.step(a)]
        fast.sync(); ieee.sync()
        diff_flags += int((ffl != ifl).sum().item()) + int((fd != idn).sum().item())
        vs_oracle += int((ffl.cpu().numpy().astype(np.uint32) != ofl).sum()) + int((ifl.cpu().numpy().astype(np.uint32) != ofl).sum())
        same = (ffl == ifl)[:, 0]
        worst = max(worst, float((fo - io).abs()[same][:, :, :34].max().item()))
        _, fi, fei = fast.get_state()
        _, ii, iei = ieee.get_state()
        diff_ints += int((fi != ii).sum()) + int((fei != iei).sum())
        n_done += int(od.sum())
        f, i, ei = ora.get_state()
        f32 = util.round_state_f32(f)
        for e in (ora, fast, ieee):
            e.set_state(f32, i, ei)
    print("fast-math vs IEEE build: flag diffs", diff_flags, "int-state diffs", diff_ints, "vs oracle", vs_oracle,
          "max |obs_fast - obs_ieee| (state block)", worst, "episodes", n_done)
    fast.close(); ieee.close()
    assert diff_flags == 0 and diff_ints == 0 and vs_oracle == 0 and worst < 5e-6 and n_done > 100


@pytest.mark.timeout(900)
def test_culling_is_result_neutral(descs):
    """The lidar's beam windows (pgd_observe.h beam_window: which beams can reach a body) and the line-free strips of straight lanes
    (pgd_upload_maps: when the line / sidewalk test can be skipped) are pure culling: a build with both switched off -- every body
    tested against every beam (-DPGD_WINDOW_SLACK=1e6f), every agent's box against the boxes of its cells every step
    (-DPGD_NO_STRIP) -- must hand out the same bits.  Free-running, the same actions on both sides: single-agent envs with every
    traffic vehicle driving (respawn mode, 240 beams) and the 40-slot roundabout (72 beams, four-wave observation)."""
    import torch
    from pgdrive_amd import build, engine
    td = tempfile.mkdtemp(prefix="pgd_nocull_")
    lib = os.path.join(td, "libpgdrive_hip_nocull.so")
    subprocess.check_call([build.hipcc(), "--offload-arch=gfx950", *build.OPT, "-std=c++17", *build.FAST_FP, "-shared", "-fPIC",
                           "-DPGD_WINDOW_SLACK=1e6f", "-DPGD_NO_STRIP", "-o", lib, build.SRC])
    L_all = engine.load_library(path=lib)
    cases = []
    mb, sb = util.make_banks(descs, n_maps=8, traffic_mode="respawn")
    cases.append(("respawn traffic", _abi.make_config(192, num_agents=1, num_traffic=16, num_lasers=240, auto_reset=1, seed=4), mb, sb, 1, 200))
    mb2, sb2 = util.make_banks(descs, n_maps=8)
    cases.append(("swerving ego", _abi.make_config(192, num_agents=1, num_traffic=16, num_lasers=240, auto_reset=1, seed=5), mb2, sb2, 1, 300))
    d, mmb, msb = util.make_marl_banks(num_agents=40, capacity=40, kind="roundabout")
    cases.append(("40 slots", util.marl_config(64, msb, horizon=150, seed=3), mmb, msb, msb.A, 250))
    for name, cfg, mb_, sb_, A, steps in cases:
        n = cfg.num_envs
        a_eng = engine.Engine(cfg, mb_, sb_)
        b_eng = engine.Engine(cfg, mb_, sb_, lib=L_all)
        ids = np.arange(n) % 8
        assert torch.equal(a_eng.reset(ids), b_eng.reset(ids))
        rng = np.random.default_rng(12)
        hits = lines = 0
        for t in range(steps):
            act = util.driving_actions(rng, n) if A == 1 else util.marl_actions(rng, n, A)
            if name == "swerving ego":  # crosses lane lines and leaves the road: the single-agent line / sidewalk test
                act[:, 0, 0] = np.clip(0.6 * np.sin(0.07 * t + np.arange(n)) + rng.normal(0, 0.1, size=n), -1, 1)
            a = torch.from_numpy(act).to(a_eng.device)
            ra = [x.clone() for x in a_eng.step(a)]
            rb = [x.clone() for x in b_eng.step(a)]
            a_eng.sync(); b_eng.sync()
            for x, y, what in zip(ra, rb, ("obs", "reward", "done", "flags")):
                assert torch.equal(x, y), "%s: %s differs at step %d" % (name, what, t)
            hits += int((ra[0][..., -cfg.num_lasers:] < 1.0).sum().item())
            lines += int(((ra[3].to(torch.int64) & (_abi.F_ON_BROKEN | _abi.F_ON_WHITE | _abi.F_ON_YELLOW | _abi.F_CRASH_SIDEWALK)) != 0).sum().item())
        print("culling off vs on, %s: %d steps x %d envs bit-identical (%d beam hits, %d line / sidewalk contacts on the way)" % (name, steps, n, hits, lines))
        assert hits > 1000 and (lines > 10 or name == "respawn traffic")  # (dense traffic ends an episode within a few steps: no line is reached)
        a_eng.close(); b_eng.close()
