"""Seeded host tables vs the reference's own RNG plumbing (goldens from oracle/gen_golden.py:gen_seeding) and the
traffic spawn rule (manager/traffic_manager.py:239-290)."""
import json
import math
import os

import numpy as np

from pgdrive_amd import mapdata, scenario

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gold():
    with open(os.path.join(GOLD, "scenes_v0.json")) as f:
        return json.load(f)


def test_get_np_random_matches_reference():
    """utils/random_utils.py:14-50: sha512-hashed RandomState seeding."""
    for row in _gold()["np_random"]:
        r = scenario.get_np_random(row["seed"])
        assert [int(r.randint(0, 65536)) for _ in range(4)] == row["randint"]
        assert abs(float(r.uniform()) - row["uniform"]) < 1e-15


def test_vehicle_parameter_sampling_matches_reference():
    """BaseRunnable.sample_parameters + ParameterSpace (base_runnable.py:81-88, utils/space.py:152-255)."""
    for row in _gold()["vehicle_params"]:
        p = scenario.sample_vehicle_params(row["vtype"], row["seed"])
        assert abs(p["max_engine_force"] - row["max_engine_force"]) < 1e-4
        assert abs(p["max_brake_force"] - row["max_brake_force"]) < 1e-4
        assert abs(math.degrees(p["max_steer"]) - row["max_steering"]) < 1e-4
        assert abs(p["friction"] - row["wheel_friction"]) < 1e-6 and p["max_speed"] == row["max_speed"]


def test_traffic_spawn_rule(descs):
    """floor(floor(sum len / 10 m) * density) vehicles per block on a 10 m grid, types from [.2,.3,.3,.2,0]."""
    d = descs[0]
    groups = scenario.propose_traffic(d, d["seed"], 0.1)
    assert len(groups) == len(d["blocks"]) - 1
    for g, b in zip(groups, d["blocks"][1:]):
        total = sum(d["lanes"][l]["length"] for lanes in b["spawn_lanes"] for l in lanes)
        assert len(g["vehicles"]) == int(math.floor(math.floor(total / 10) * 0.1))
        for v in g["vehicles"]:
            assert v["long"] % 10 == 0 and v["long"] < d["lanes"][v["lane"]]["length"]
            assert v["vtype"] in ("s", "m", "l", "xl")
    again = scenario.propose_traffic(d, d["seed"], 0.1)
    assert again == groups  # same seed -> same traffic (test_random_engine.py:6-188)
    assert scenario.propose_traffic(d, d["seed"], 0.0) == []


def test_scenario_records(descs):
    sb = scenario.ScenarioBank(descs[:5], [d["seed"] for d in descs[:5]], num_agents=1, num_traffic=16)
    assert sb.spawns.shape == (5 * 17, )
    for k, d in enumerate(descs[:5]):
        sp = sb.spawns[k * 17:(k + 1) * 17]
        ego = sp[0]
        assert ego["lane"] == 0 and ego["group"] == -1
        assert abs(ego["x"] - 5.0) < 1e-6 and abs(ego["y"]) < 1e-6 and ego["heading"] == 0.0  # spawn long 5 on ('>','>>',0)
        assert 750 <= ego["max_engine_force"] <= 850 and 80 <= ego["max_brake_force"] <= 180
        n = ego["n_ckpt"]
        assert d["nodes"][ego["ckpt"][0]] == ">" and n >= 3
        # every leg of the route is a road of the map, the destination lane is the right-most lane of the last road
        for j in range(n - 1):
            r = d["roads"][ego["ckpt_road"][j]]
            assert r["frm"] == ego["ckpt"][j] and r["to"] == ego["ckpt"][j + 1]
        last = d["roads"][ego["ckpt_road"][n - 2]]
        assert ego["dest_lane"] == last["first_lane"] + last["n_lanes"] - 1
        used = sp[1:][sp[1:]["lane"] >= 0]
        assert (used["group"] >= 0).all() and (used["timer0"] >= 0).all() and (used["timer0"] < 50).all()
        assert sb.scenarios[k]["n_groups"] == len(d["blocks"]) - 1


def test_empty_and_capped_slots(descs):
    """density 0 -> no traffic slots used; a tiny cap drops vehicles but keeps the RNG stream aligned."""
    sc, sp, info = scenario.build_scenario(descs[2], 0, descs[2]["seed"], num_traffic=16, density=0.0)
    assert (sp[1:]["lane"] < 0).all() and info["n_traffic"] == 0
    sc2, sp2, info2 = scenario.build_scenario(descs[2], 0, descs[2]["seed"], num_traffic=2, density=0.3)
    sc3, sp3, info3 = scenario.build_scenario(descs[2], 0, descs[2]["seed"], num_traffic=40, density=0.3)
    assert info2["dropped"] > 0 and info3["dropped"] == 0
    assert (sp2[1:3]["lane"] == sp3[1:3]["lane"]).all() and (sp2[1:3]["max_engine_force"] == sp3[1:3]["max_engine_force"]).all()


def test_traffic_proposals_match_reference_manager():
    """The reference's own TrafficManager._create_vehicles_once / _create_respawn_vehicles (traffic_manager.py:188-309), run
    on its block objects with recorders in place of spawn_object / IDMPolicy (oracle/gen_golden.py:gen_traffic): same
    lanes, longitudes, vehicle types and policy seeds, vehicle by vehicle, for trigger and respawn modes."""
    from pgdrive_amd import bank
    with open(os.path.join(GOLD, "traffic_v0.json")) as f:
        gold = json.load(f)["maps"]
    descs = {d["seed"]: d for d in bank.get_descriptions([g["seed"] for g in gold])}
    n = 0
    for g in gold:
        d = descs[g["seed"]]
        rl = mapdata.road_lookup(d)
        for density in (0.1, 0.3):
            mine = scenario.propose_traffic(d, d["seed"], density)
            ref = g["trigger_%g" % density]
            assert len(mine) == len(ref)
            for a, b in zip(mine, ref):
                assert a["trigger_road"] == rl[tuple(b["trigger"])]
                assert [[v["lane"], v["long"], v["vtype"], v["policy_seed"]] for v in a["vehicles"]] == b["vehicles"]
                n += len(b["vehicles"])
            mine = scenario.propose_respawn_traffic(d, d["seed"], density)
            ref = g["respawn_%g" % density]
            assert [[v["lane"], v["long"], v["vtype"], v["policy_seed"]] for v in mine[0]["vehicles"]] == ref
            n += len(ref)
    assert n > 300


def test_respawn_mode_and_auto_termination_scenarios(descs):
    d = descs[0]
    sc, sp, info = scenario.build_scenario(d, 0, d["seed"], 1, 16, 0.1, traffic_mode="respawn", auto_termination=True)
    assert sc["n_groups"] == 0 and sc["max_steps"] == 250 * len(d["blocks"])
    assert (sp["group"][1:][sp["lane"][1:] >= 0] == -1).all() and info["n_traffic"] == 16
    sc, sp, info = scenario.build_scenario(d, 0, d["seed"], 1, 16, 0.1, traffic_mode="hybrid")
    sc2, sp2, _ = scenario.build_scenario(d, 0, d["seed"], 1, 16, 0.1)
    assert sc.tobytes() == sc2.tobytes() and sp.tobytes() == sp2.tobytes() and sc["max_steps"] == 0
    sc3, sp3, _ = scenario.build_scenario(d, 0, d["seed"], 1, 16, 0.1, traffic_seed=7)
    assert sp3.tobytes() != sp2.tobytes() and sp3[0].tobytes() == sp2[0].tobytes()  # only the traffic moves
    import pytest
    with pytest.raises(ValueError):
        scenario.build_scenario(d, 0, d["seed"], 1, 16, 0.1, traffic_mode="nope")


def test_object_proposals_match_reference_manager():
    """TrafficObjectManager.reset (object_manager.py:40-124) run on the reference's block objects with a recording
    spawn_object (oracle/gen_golden.py:gen_objects): same scenes, lanes, longitudes, laterals, world positions; the
    traffic manager that follows skips the accident lanes and has its type stream advanced by the broken-down vehicles."""
    from pgdrive_amd import mapgen
    with open(os.path.join(GOLD, "objects_v0.json")) as f:
        cases = json.load(f)["cases"]
    cls_name = dict(cone="TrafficCone", warning="TrafficWarning", barrier="TrafficBarrier")
    n_obj = n_veh = 0
    for c in cases:
        d = mapgen.generate_map(c["seed"], **c["kw"])
        objs, acc, n_draws = scenario.propose_objects(d, c["seed"], c["prob"])
        assert acc == c["accident_lanes"]
        assert len(objs) == len(c["objects"])
        res = scenario.propose_traffic(d, c["seed"], 0.2, skip_lanes=set(acc), type_draws=n_draws)
        groups, pre_types = res if n_draws else (res, [])
        for o, r in zip(objs, c["objects"]):
            assert o["lane"] == r[1] and abs(o["long"] - r[2]) < 1e-9 and abs(o["lat"] - r[3]) < 1e-9
            if o["cls"] == "vehicle":
                assert r[0] == "vehicle" and pre_types[o["type_draw"]] == r[4]
                n_veh += 1
            else:
                assert cls_name[o["cls"]] == r[0]
                L = d["lanes"][o["lane"]]
                x, y = mapdata.lane_position(L, o["long"], o["lat"])
                assert abs(x - r[4]) < 1e-9 and abs(y - r[5]) < 1e-9
                assert abs(mapdata.lane_heading_at(L, o["long"]) - r[6]) < 1e-12
            n_obj += 1
        mine = [[[v["lane"], v["long"], v["vtype"], v["policy_seed"]] for v in g["vehicles"]] for g in groups]
        assert mine == c["traffic_0_2"]
    assert n_obj > 250 and n_veh >= 3


def test_random_agent_model_type_draw(descs):
    """AgentManager._get_vehicles (agent_manager.py:63-73): one uniform vehicle-type draw per episode from the manager's
    stream; the ego's record then carries that type's dimensions."""
    with open(os.path.join(GOLD, "traffic_v0.json")) as f:
        types = json.load(f)["random_agent_types"]
    from pgdrive_amd import bank
    seen = set()
    for seed, vt in types.items():
        d = bank.get_descriptions([int(seed)])[0]
        sc, sp, info = scenario.build_scenario(d, 0, int(seed), 1, 4, 0.1, random_agent_model=True)
        assert abs(float(sp[0]["length"]) - scenario.VEHICLE_TYPES[vt]["length"]) < 1e-6, (seed, vt)
        assert abs(float(sp[0]["width"]) - scenario.VEHICLE_TYPES[vt]["width"]) < 1e-6
        seen.add(vt)
    assert len(seen) >= 4


def test_spawn_lane_index_and_destination_node_overrides():
    """vehicle_config.spawn_lane_index / destination_node (pgdrive_env.py:76-80): the ego starts on the named lane and its
    route ends at the named node (Navigation.set_route, navigation.py:123-148)."""
    from pgdrive_amd import bank, scenario
    descs = bank.get_descriptions([1000, 1003])
    base = scenario.ScenarioBank(descs, [1000, 1003], num_agents=1, num_traffic=4)
    for m, d in enumerate(descs):
        # default spawn = lane 0 of '>' -> '>>'
        lane0 = scenario.resolve_lane_index(d, (">", ">>", 0))
        assert base.spawns[m * base.V]["lane"] == lane0
    # spawn on lane 2 of the first road, destination = end of the first block's straight
    dest = descs[0]["nodes"][4]  # '>>>'
    sb = scenario.ScenarioBank(descs[:1], [1000], num_agents=1, num_traffic=4, spawn_lane_index=(">", ">>", 2),
                               destination_node=dest)
    ego = sb.spawns[0]
    assert ego["lane"] == scenario.resolve_lane_index(descs[0], (">", ">>", 2))
    n = int(ego["n_ckpt"])
    assert descs[0]["nodes"][int(ego["ckpt"][n - 1])] == dest
    # lateral position: lane 2 is two lane widths to the right of lane 0
    e0 = base.spawns[0]
    assert abs(np.hypot(ego["x"] - e0["x"], ego["y"] - e0["y"]) - 2 * descs[0]["lane_width"]) < 1e-4
    import pytest
    with pytest.raises(KeyError):
        scenario.ScenarioBank(descs[:1], [1000], num_agents=1, num_traffic=4, spawn_lane_index=(">", ">>", 7))
    with pytest.raises(KeyError):
        scenario.ScenarioBank(descs[:1], [1000], num_agents=1, num_traffic=4, destination_node="nowhere")
