"""Multi-agent roundabout, CPU side: map boxes + spawn-slot table vs reference goldens, and the oracle's episode protocol
(multi_agent_pgdrive.py:109-213) — no GPU needed."""
import json
import os

import numpy as np

from pgdrive_amd import _abi, bank, mapdata, scenario
from tests import util

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _desc():
    return bank.load_descriptions(bank.MA_ROUNDABOUT_BANK)[0]


def test_roundabout_boxes_match_reference():
    d = _desc()
    ref = np.load(os.path.join(GOLD, "boxes_ma_roundabout.npz"))["boxes"]
    mine = mapdata.build_boxes(d)
    assert ref.shape == mine.shape == (704, 7)
    assert (ref[:, 0] == mine[:, 0]).all() and (ref[:, 6] == mine[:, 6]).all()
    assert np.abs(ref[:, [1, 2, 4, 5]] - mine[:, [1, 2, 4, 5]]).max() < 1e-9
    assert len(d["lanes"]) == 52


def test_spawn_slots_match_reference():
    """SpawnManager slot table (spawn_manager.py:114-155) on MARoundaboutConfig.spawn_roads (marl_inout_roundabout.py:15-28)."""
    with open(os.path.join(GOLD, "scenes_v0.json")) as f:
        g = json.load(f)
    d = _desc()
    roads = scenario.roundabout_spawn_roads(d)
    slots, safe = scenario.spawn_slots(d, roads)
    assert len(slots) == g["marl_capacity"] == 48 and len(safe) == 8
    for mine, ref in zip(slots, g["marl_slots"]):
        assert [d["nodes"][mine["road"][0]], d["nodes"][mine["road"][1]]] == ref["road"]
        lane = d["lanes"][mine["lane"]]
        assert lane["index"] == ref["lane_idx"] and mine["long"] == ref["long"]
        x, y = mapdata.lane_position(lane, mine["long"], 0.0)
        assert abs(x - ref["x"]) < 1e-9 and abs(y - ref["y"]) < 1e-9
        assert abs(mapdata.lane_heading_at(lane, mine["long"]) - ref["heading"]) < 1e-12
    dests = [d["nodes"][scenario.neg_road(d, *r)[1]] for r in roads]
    assert dests == g["marl_dest_nodes"]
    import pytest
    with pytest.raises(ValueError, match="Too many agents"):
        scenario.MarlScenarioBank(d, num_agents=49, n_variants=1)


def test_infinite_agents_fill_every_spawn_slot_in_order():
    """num_agents = -1 ("as many vehicles as possible", base_env.py:25): SpawnManager.reset takes every slot in slot order instead of
    a draw (spawn_manager.py:76-81), and AgentManager.allow_respawn stops counting agents (agent_manager.py:316-323)."""
    d = _desc()
    slots, safe = scenario.spawn_slots(d, scenario.roundabout_spawn_roads(d))
    sb = scenario.MarlScenarioBank(d, num_agents=-1, n_variants=3, seed=4)
    assert sb.infinite and sb.num_agents == sb.A == len(slots) == 48
    per = sb.spawns.reshape(3, -1)
    lo = abs(scenario.RESPAWN_REGION_LONGITUDE - scenario.MAX_VEHICLE_LENGTH)
    la = abs(scenario.RESPAWN_REGION_LATERAL - scenario.MAX_VEHICLE_WIDTH)
    for v in range(3):
        assert (per[v]["lane"][:48] == np.array([c["lane"] for c in slots])).all()  # slot order, every variant
        # (jittered inside the slot, spawn_manager.py:157-166)
        for a, c in enumerate(slots):
            x, y = mapdata.lane_position(d["lanes"][c["lane"]], c["long"], 0.0)
            assert np.hypot(per[v]["x"][a] - x, per[v]["y"][a] - y) <= np.hypot(0.5 * lo, 0.5 * la) + 1e-4
    big = scenario.MarlScenarioBank(d, num_agents=-1, capacity=60, n_variants=1)  # room to grow beyond the initial 48
    assert big.A == 60 and big.num_agents == 48 and (big.spawns["lane"][48:60] == -1).all()
    # fewer seats than spawn slots (`max_agents`): the first seats' worth of slots, in slot order (this used to raise)
    small = scenario.MarlScenarioBank(d, num_agents=-1, capacity=40, n_variants=2)
    assert small.infinite and small.A == small.num_agents == 40
    assert (small.spawns.reshape(2, -1)[1]["lane"][:40] == np.array([c["lane"] for c in slots[:40]])).all()
    import pytest
    with pytest.raises(ValueError, match="Too many agents"):
        scenario.MarlScenarioBank(d, num_agents=50, capacity=50, n_variants=1)  # 48 spawn slots


def test_target_vehicle_configs_pin_the_named_agents():
    """multi_agent_pgdrive.py:96-107 + spawn_manager.py:58-69,91-100: an agent named in `target_vehicle_configs` starts where it is
    told (lane by node names, longitude, lateral, optionally its destination); the others keep their drawn slots, and the draws of
    the generator are the same with and without the override (the placement replaces the drawn one afterwards)."""
    from pgdrive_amd import bank
    d = bank.get_descriptions([0], 3, 3.5, 50, block_seq="SSS", block_num=None)[0]
    plain = scenario.MarlScenarioBank([d], num_agents=4, n_variants=2, seed=3, kind="pg")
    fixed = {k: dict(spawn_longitude=5.0 * k) for k in range(4)}
    fixed[2]["spawn_lane_index"] = (">", ">>", 1)
    ref0 = plain.spawns.reshape(2, -1)[0]
    dest_node = d["nodes"][int(ref0["ckpt"][0][int(ref0["n_ckpt"][0]) - 1])]  # a reachable node: where agent 0 is headed anyway
    mid_node = d["nodes"][int(ref0["ckpt"][0][int(ref0["n_ckpt"][0]) - 2])]   # ... and one node before it: a shorter route
    fixed[3]["destination_node"] = mid_node
    sb = scenario.MarlScenarioBank([d], num_agents=4, n_variants=2, seed=3, kind="pg", fixed=fixed)
    first = scenario.resolve_lane_index(d, (">", ">>", 0))
    per, ref = sb.spawns.reshape(2, -1), plain.spawns.reshape(2, -1)
    for v in range(2):
        for k in range(4):
            lane = d["lanes"][first + (1 if k == 2 else 0)]
            x, y = mapdata.lane_position(lane, 5.0 * k, 0.0)
            assert per[v]["lane"][k] == first + (1 if k == 2 else 0)
            assert abs(per[v]["x"][k] - x) < 1e-5 and abs(per[v]["y"][k] - y) < 1e-5
            # same vehicle parameters as without the override: the generator's stream is untouched
            assert per[v]["length"][k] == ref[v]["length"][k] and per[v]["max_engine_force"][k] == ref[v]["max_engine_force"][k]
        n = int(per[v]["n_ckpt"][3])
        assert d["nodes"][int(per[v]["ckpt"][3][n - 1])] == mid_node != dest_node
        # the respawn table behind the agent slots is the same
        assert (per[v][4:] == ref[v][4:]).all()
    import pytest
    with pytest.raises(KeyError):
        scenario.MarlScenarioBank([d], num_agents=4, n_variants=1, kind="pg", fixed={0: dict(destination_node="no such node")})
    with pytest.raises(KeyError, match="agent99"):  # a name outside agent0 .. agent3 is refused, not silently ignored
        scenario.MarlScenarioBank([d], num_agents=4, n_variants=1, kind="pg", fixed={99: dict(spawn_longitude=5.0)})


def test_spawn_roads_override():
    """`spawn_roads` of the multi-agent configs (marl_inout_roundabout.py:15-28): given as Road-like objects or node-name pairs, the
    slots, the safe respawn places and the destination list follow the given roads."""
    d = _desc()
    n = d["nodes"]

    class Road:  # what the reference's Road offers (road.py): start_node / end_node
        def __init__(self, a, b):
            self.start_node, self.end_node = a, b

    two = [Road(">>", ">>>"), (n[scenario.roundabout_spawn_roads(d)[1][0]], n[scenario.roundabout_spawn_roads(d)[1][1]])]
    sb = scenario.MarlScenarioBank(d, num_agents=6, n_variants=2, seed=0, spawn_roads=two)
    full = scenario.MarlScenarioBank(d, num_agents=6, n_variants=2, seed=0)
    assert (sb.P, sb.Dn) == (4, 2) and (full.P, full.Dn) == (8, 4)  # 2 lanes x roads safe places; one destination per road
    lanes_ok = set()
    for r in scenario.resolve_spawn_roads(d, two):
        road = d["roads"][mapdata.road_lookup(d)[r]]
        lanes_ok |= set(range(road["first_lane"], road["first_lane"] + road["n_lanes"]))
    per = sb.spawns.reshape(2, -1)
    assert set(per[0]["lane"][:6].tolist()) <= lanes_ok and set(per[0]["lane"][6:].tolist()) <= lanes_ok
    import pytest
    with pytest.raises(KeyError):
        scenario.MarlScenarioBank(d, num_agents=2, n_variants=1, spawn_roads=[(">>", "nowhere")])
    with pytest.raises(ValueError, match="Too many agents"):
        scenario.MarlScenarioBank(d, num_agents=13, n_variants=1, spawn_roads=[(">>", ">>>")])  # 2 lanes x 6 slots


def test_oracle_marl_episode_protocol():
    """delay-done queue, respawn ids, horizon, __all__ + auto-reset on the CPU oracle."""
    from oracle import orc
    d, mb, sb = util.make_marl_banks(num_agents=8, n_variants=4)
    n = 4
    cfg = util.marl_config(n, sb, horizon=80)
    o = orc.Oracle(cfg, mb, sb)
    obs = o.reset(np.arange(n) % 4)
    assert obs.shape == (n, 8, 90)
    rng = np.random.default_rng(0)
    SI, SF, EI = _abi.SI, _abi.SF, _abi.EI
    max_id = np.full(n, 7)
    n_all = 0
    prev_status = o.get_state()[1][SI["STATUS"]].copy()
    dying_age = np.zeros((n, 8), dtype=int)
    for t in range(300):
        act = util.marl_actions(rng, n, 8)
        obs, rew, done, fl = o.step(act)
        f, i, ei = o.get_state()
        st = i[SI["STATUS"]]
        report = (fl & _abi.F_REPORT) != 0
        new = (fl & _abi.F_NEW) != 0
        reset = (fl & _abi.F_RESET) != 0
        # only agents that were active at the start of the step report
        assert (report == (prev_status == _abi.ST_ACTIVE)).all()
        assert (done[~report] == 0).all() and (rew[~report] == 0).all()
        # a finished agent either leaves at once (arrival) or becomes a static dying body for delay_done steps
        fin = report & (done == 1) & ~reset
        assert ((st[fin] == _abi.ST_DYING) | (st[fin] == _abi.ST_EMPTY)).all()
        assert (i[SI["TIMER"]][fin & (st == _abi.ST_DYING)] == 25).all()
        dying_age = np.where(st == _abi.ST_DYING, dying_age + 1, 0)
        assert dying_age.max() <= 25
        # newcomers carry fresh, increasing ids; NEXT_AGENT counts them
        ids = f[SF["AGENT_ID"]].astype(int)
        for e in range(n):
            if reset[e].any():
                assert (ei[EI["EP_STEPS"], e] == 0) and ei[EI["NEXT_AGENT"], e] == 8
                max_id[e] = 7
                n_all += 1
                continue
            for s in np.nonzero(new[e])[0]:
                assert ids[e, s] > max_id[e]
                max_id[e] = ids[e, s]
            assert ei[EI["NEXT_AGENT"], e] == max_id[e] + 1
            alive = ((st[e] == _abi.ST_ACTIVE) | (st[e] == _abi.ST_DYING)).sum()
            assert alive <= 8
        assert np.isfinite(obs).all() and obs.min() >= 0 and obs.max() <= 1
        prev_status = st.copy()
    assert n_all >= n  # every env finished at least one episode (horizon 80 -> __all__ soon after)


def test_intersection_map_and_slots_match_reference():
    """MAIntersectionMap (marl_intersection.py:29-54) from our generator: boxes vs the reference's recorded Bullet boxes,
    spawn slot table and destinations vs MAIntersectionConfig.spawn_roads (marl_intersection.py:14-20)."""
    from pgdrive_amd import mapgen
    d = mapgen.generate_ma_intersection()
    ref = np.load(os.path.join(GOLD, "boxes_ma_intersection.npz"))["boxes"]
    mine = mapdata.build_boxes(d)
    assert ref.shape == mine.shape
    assert (ref[:, 0] == mine[:, 0]).all() and (ref[:, 6] == mine[:, 6]).all()
    assert np.abs(ref[:, [1, 2, 4, 5]] - mine[:, [1, 2, 4, 5]]).max() < 1e-9
    with open(os.path.join(GOLD, "marl_intersection_v0.json")) as f:
        g = json.load(f)
    roads = scenario.intersection_spawn_roads(d)
    slots, safe = scenario.spawn_slots(d, roads)
    assert len(slots) == g["capacity"] == 48 and len(safe) == 8
    for mine_s, r in zip(slots, g["slots"]):
        assert [d["nodes"][mine_s["road"][0]], d["nodes"][mine_s["road"][1]]] == r["road"]
        lane = d["lanes"][mine_s["lane"]]
        assert lane["index"] == r["lane_idx"] and mine_s["long"] == r["long"]
        x, y = mapdata.lane_position(lane, mine_s["long"], 0.0)
        assert abs(x - r["x"]) < 1e-9 and abs(y - r["y"]) < 1e-9
        assert abs(mapdata.lane_heading_at(lane, mine_s["long"]) - r["heading"]) < 1e-12
    assert [d["nodes"][scenario.neg_road(d, *r)[1]] for r in roads] == g["dest_nodes"]
    # every (safe place, destination) pair has a route, u-turn destinations included
    sb = scenario.MarlScenarioBank(d, num_agents=30, n_variants=2, kind="intersection")
    assert sb.P == 8 and sb.Dn == 4 and (sb.spawns["n_ckpt"][sb.spawns["lane"] >= 0] >= 2).all()


def test_oracle_intersection_episode_runs():
    from oracle import orc
    d, mb, sb = util.make_marl_banks(num_agents=30, n_variants=2, kind="intersection")
    cfg = util.marl_config(2, sb, horizon=60)
    o = orc.Oracle(cfg, mb, sb)
    obs = o.reset(np.arange(2) % 2)
    assert obs.shape == (2, 30, 90)
    rng = np.random.default_rng(0)
    n_new = n_all = n_arrive = 0
    for t in range(200):
        obs, rew, done, fl = o.step(util.marl_actions(rng, 2, 30))
        n_new += int(((fl & _abi.F_NEW) != 0).sum())
        n_all += int(((fl & _abi.F_ALL_DONE) != 0).any(axis=1).sum())
        n_arrive += int(((fl & _abi.F_ARRIVE) != 0).sum())
        assert np.isfinite(obs).all() and obs.min() >= 0.0 and obs.max() <= 1.0
    assert n_new > 10 and n_all >= 2
    o.close()


def test_neighbour_state_rows_layout():
    """PGD_MA_OTHERS_STATE (LidarStateObservationMARound, marl_inout_roundabout.py:72-105): row = state | num_others x state |
    lidar; a neighbour's row equals the state block of that neighbour's own observation; absent ranks are zero."""
    from oracle import orc
    d, mb, sb = util.make_marl_banks(num_agents=8, capacity=8, kind="roundabout")
    n_envs, NO = 4, 3
    cfg = util.marl_config(n_envs, sb, num_others=NO, others_state=True)
    SL = 2 + 6 + 10
    assert _abi.obs_dim(cfg) == SL + NO * SL + 72
    ora = orc.Oracle(cfg, mb, sb)
    ora.reset(np.arange(n_envs) % 4)
    rng = np.random.default_rng(0)
    for t in range(30):
        obs = ora.step(util.marl_actions(rng, n_envs, sb.A))[0]
    f, i, ei = ora.get_state()
    checked = zeros = 0
    for e in range(n_envs):
        act = np.nonzero(i[_abi.SI["STATUS"], e, :sb.A] == _abi.ST_ACTIVE)[0]
        for a in act:
            row = obs[e, a]
            for r in range(NO):
                blk = row[SL + r * SL:SL + (r + 1) * SL]
                if not blk.any():
                    zeros += 1
                    continue
                # it must be the state block of exactly one other active agent's own row
                hits = [b for b in act if b != a and np.allclose(obs[e, b, :SL], blk, atol=1e-12)]
                others_dying = (i[_abi.SI["STATUS"], e, :sb.A] == _abi.ST_DYING).any()
                assert hits or others_dying, (e, a, r)
                checked += bool(hits)
    assert checked > 10


def test_generic_multi_agent_scenarios_over_generated_maps():
    """MultiAgentPGDrive itself (multi_agent_pgdrive.py:12-55): spawn road '>>' -> '>>>' of generated maps = 5 slots x 3 lanes,
    destinations by Navigation's seeded default, the same respawn-table shape on every map; oracle episode runs."""
    from oracle import orc
    descs = bank.get_descriptions([3, 4, 5])
    sb = scenario.MarlScenarioBank(descs, 15, n_variants=2, seed=1, kind="pg")
    assert sb.P == 3 and sb.Dn == 1 and sb.B == 0 and len(sb.scenarios) == 6  # respawn only into the safe first slot of a lane
    assert sorted(set(int(m) for m in sb.scenarios["map"])) == [0, 1, 2]
    with __import__("pytest").raises(ValueError):
        scenario.MarlScenarioBank(descs[:1], 16, n_variants=1, kind="pg")  # only 15 slots
    mb = mapdata.MapBank(descs, truncate_succ=True)
    n_envs = 6
    cfg = util.marl_config(n_envs, sb, horizon=200)
    ora = orc.Oracle(cfg, mb, sb)
    obs = ora.reset(np.arange(n_envs))
    assert obs.shape == (n_envs, 15, 18 + 72) and np.isfinite(obs).all()
    rng = np.random.default_rng(0)
    new = 0
    for t in range(260):
        o, r, d, fl = ora.step(util.marl_actions(rng, n_envs, 15))
        new += int(((fl & _abi.F_NEW) != 0).sum())
        assert np.isfinite(o).all() and o.min() >= 0.0 and o.max() <= 1.0
    assert new > 15 * n_envs  # respawns happened after the initial placement
