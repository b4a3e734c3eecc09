"""Host-side episode set-up: ego spawn, destination + route, traffic spawn table (one *scenario* per map seed).

Restates, on flat map descriptions, what the reference does once per `env.reset()`:

* seeding                     pgdrive/utils/random_utils.py:14-50, base_class/randomizable.py:4-21
* ego spawn                   pgdrive/envs/pgdrive_env.py:77-79, component/vehicle/base_vehicle.py:292-339
* destination + BFS route     component/vehicle_module/navigation.py:99-153
* traffic spawn (trigger mode) manager/traffic_manager.py:239-290,311-314
* vehicle parameter sampling  base_class/base_runnable.py:81-88, utils/space.py:219-255, vehicle/vehicle_type.py:7-86
* IDM overtake timer          policy/idm_policy.py:185

The result is packed into `pgd_scenario` / `pgd_spawn` records (include/pgdrive_hip.h) and uploaded once.
"""
import hashlib
import math
import struct

import numpy as np

from . import mapdata

MAX_CKPT = mapdata.MAX_CKPT
MAX_GROUPS = 16

SPAWN_DT = np.dtype(
    [
        ("x", "<f4"), ("y", "<f4"), ("heading", "<f4"), ("length", "<f4"), ("width", "<f4"), ("wheelbase", "<f4"),
        ("mass", "<f4"), ("max_engine_force", "<f4"), ("max_brake_force", "<f4"), ("friction", "<f4"),
        ("max_steer", "<f4"), ("max_speed", "<f4"), ("lane", "<i2"), ("group", "<i2"), ("n_ckpt", "<i2"),
        ("timer0", "<i2"), ("dest_lane", "<i2"), ("kind", "<i2"), ("aux", "<i2"), ("pad", "<i2"), ("ckpt", "<i2", (MAX_CKPT, )),
        ("ckpt_road", "<i2", (MAX_CKPT, ))
    ]
)
SCEN_DT = np.dtype([("map", "<i4"), ("n_groups", "<i4"), ("trigger_road", "<i2", (MAX_GROUPS, )), ("max_steps", "<i4"),
                    ("aux", "<i4")])
assert SPAWN_DT.itemsize == 64 + 4 * MAX_CKPT and SCEN_DT.itemsize == 16 + 2 * MAX_GROUPS

# vehicle_type.py:7-74 (L, W, front+rear wheelbase, mass) and utils/space.py:219-255
# parameter tuples are (first, second) positional args of the reference's BoxSpace namedtuple("max min").
VEHICLE_TYPES = {
    "default": dict(length=4.51, width=1.852, wheelbase=1.05234 + 1.4166, mass=1100.0, friction=0.9,
                    engine=(750, 850), brake=(80, 180), max_steering=40.0, max_speed=80.0),
    "s": dict(length=4.25, width=1.7, wheelbase=1.4126 + 1.07, mass=800.0, friction=0.9, engine=(350, 550),
              brake=(35, 80), max_steering=50.0, max_speed=80.0),
    "m": dict(length=4.4, width=1.85, wheelbase=1.285 + 1.203, mass=1200.0, friction=0.75, engine=(650, 850),
              brake=(60, 150), max_steering=45.0, max_speed=80.0),
    "l": dict(length=4.5, width=1.86, wheelbase=1.391 + 1.10751, mass=1300.0, friction=0.8, engine=(450, 650),
              brake=(60, 120), max_steering=40.0, max_speed=80.0),
    "xl": dict(length=5.8, width=2.3, wheelbase=1.726 + 1.075, mass=1600.0, friction=0.7, engine=(500, 700),
               brake=(50, 100), max_steering=35.0, max_speed=80.0),
}
TYPE_KEYS = ["s", "m", "l", "xl", "default"]  # dict order of vehicle_type (vehicle_type.py:78)
TRAFFIC_TYPE_PROB = [0.2, 0.3, 0.3, 0.2, 0]  # traffic_manager.py:313
VEHICLE_GAP = 10  # traffic_manager.py:31
MAX_RAND_INT = 65536  # randomizable.py:10


def get_np_random(seed):
    """RandomState seeded with the first 8 bytes of sha512(str(seed)) (random_utils.py:14-50, 60-83)."""
    seed = int(seed) % 2**64
    h = hashlib.sha512(str(seed).encode("utf8")).digest()[:8]
    b = h + b"\0" * (4 - len(h) % 4)  # the reference always pads (even when already aligned)
    words = struct.unpack("{}I".format(len(b) // 4), b)
    big = sum(2**(32 * i) * w for i, w in enumerate(words))
    ints = []
    if big == 0:
        ints = [0]
    while big > 0:
        big, mod = divmod(big, 2**32)
        ints.append(mod)
    rng = np.random.RandomState()
    rng.seed(ints)
    return rng


def sample_vehicle_params(vtype, object_seed):
    """BaseRunnable.sample_parameters (base_runnable.py:81-88): every Box of the ParameterSpace is seeded with the same
    integer, so each draws the same first uniform; values are cast to float32 (Box dtype)."""
    t = VEHICLE_TYPES[vtype]
    rng = get_np_random(object_seed)
    ps_seed = rng.randint(low=0, high=int(1e6))

    def box(first, second):  # BoxSpace(first, second) -> max=first, min=second -> Box(low=second, high=first)
        r = get_np_random(ps_seed)
        return float(np.float32(r.uniform(low=np.float32(second), high=np.float32(first))))

    return dict(
        length=t["length"], width=t["width"], wheelbase=t["wheelbase"], mass=t["mass"], friction=t["friction"],
        max_engine_force=box(*t["engine"]), max_brake_force=box(*t["brake"]),
        max_steer=math.radians(t["max_steering"]), max_speed=t["max_speed"], vtype=vtype,
    )


def choose_destination(desc, seed, start_node, negative=False):
    """Navigation.update (navigation.py:99-121): a socket of the last block, chosen with get_np_random(seed)."""
    block = desc["blocks"][0] if negative else desc["blocks"][-1]
    sockets = list(block["sockets"])
    rng = get_np_random(seed)
    socket = sockets[rng.choice(len(sockets))]
    while True:
        is_socket_node = start_node in (socket["pos"][0], socket["pos"][1], socket["neg"][0], socket["neg"][1])
        if not is_socket_node or len(sockets) == 1:
            break
        sockets.remove(socket)
        if len(sockets) == 0:
            raise ValueError("Can not set a destination!")
    return socket["neg"][1] if negative else socket["pos"][1]


def make_route(desc, lane_id, final_node):
    """Navigation.set_route (navigation.py:123-148) -> (ckpt nodes, road per leg, final lane id)."""
    rl = mapdata.road_lookup(desc)
    lane = desc["lanes"][lane_id]
    road = desc["roads"][lane["road"]]
    ckpt = mapdata.shortest_path(desc, road["frm"], final_node)
    if len(ckpt) <= 2:
        ckpt = [road["frm"], road["to"]]
        single = True
    else:
        single = False
    roads = [rl[(ckpt[k], ckpt[k + 1])] for k in range(len(ckpt) - 1)]
    final_road = desc["roads"][roads[-1]]
    final_lane = final_road["first_lane"] + final_road["n_lanes"] - 1  # final_lanes[-1]
    return ckpt, roads, final_lane, single


def _fill_route(rec, desc, lane_id, final_node):
    ckpt, roads, final_lane, single = make_route(desc, lane_id, final_node)
    if len(ckpt) > MAX_CKPT:
        raise ValueError("route with %d nodes exceeds PGD_MAX_CKPT" % len(ckpt))
    rec["n_ckpt"] = len(ckpt)
    rec["ckpt"][:] = -1
    rec["ckpt_road"][:] = -1
    rec["ckpt"][:len(ckpt)] = ckpt
    rec["ckpt_road"][:len(roads)] = roads
    rec["dest_lane"] = final_lane
    return single


def _fill_vehicle(rec, desc, lane_id, longitude, lateral, params):
    l = desc["lanes"][lane_id]
    x, y = mapdata.lane_position(l, longitude, lateral)
    rec["x"], rec["y"] = x, y
    rec["heading"] = mapdata.lane_heading_at(l, longitude)
    for k in ("length", "width", "wheelbase", "mass", "max_engine_force", "max_brake_force", "friction", "max_steer",
              "max_speed"):
        rec[k] = params[k]
    rec["lane"] = lane_id


# ----------------------------------------------------------------------------------------------------------------------
# traffic objects (manager/object_manager.py, component/static_object/traffic_object.py)
# ----------------------------------------------------------------------------------------------------------------------
OBJ_VEHICLE, OBJ_CYLINDER, OBJ_BOX = 0, 1, 2  # pgd_spawn.kind
GROUP_NEVER = -2  # PGD_GROUP_NEVER
CONE_RADIUS, WARNING_RADIUS = 0.25, 0.5  # traffic_object.py:40,60
BARRIER_LENGTH, BARRIER_WIDTH = 2.0, 0.3  # traffic_object.py:84-85
ALERT_DIST, ACCIDENT_AREA_LEN, CONE_LONGITUDE, CONE_LATERAL, PROHIBIT_SCENE_PROB = 10, 10, 2, 1, 0.67  # object_manager.py:18-26


def propose_objects(desc, seed, accident_prob):
    """TrafficObjectManager.reset (object_manager.py:40-124) with its own RNG stream (BaseManager seeds it with the episode
    seed): per Straight / Curve / ramp block an accident with probability `accident_prob`: a cone-fenced construction
    site at a lane end ("prohibit scene"), a broken-down vehicle with a warning tripod 10 m behind it, or a barrier.
    Returns (objects in spawn order, accident lane ids, number of vehicle-type draws taken from the TRAFFIC manager's RNG).
    Object = dict(cls, lane, long, lat[, vtype])."""
    objs, accident_lanes, n_type_draws = [], [], 0
    if abs(accident_prob) < 1e-2:
        return objs, accident_lanes, n_type_draws
    rng = get_np_random(seed)
    nodes = desc["nodes"]
    rl = mapdata.road_lookup(desc)

    def lanes_of(road):
        r = desc["roads"][rl[road]]
        return list(range(r["first_lane"], r["first_lane"] + r["n_lanes"]))

    for bi, block in enumerate(desc["blocks"]):
        if block["id"] not in ("S", "C", "r", "R"):
            continue
        if rng.rand() > accident_prob:
            continue
        n00 = nodes.index("%d%s0_0_" % (bi, block["id"]))
        road_1 = (block["trigger_road"][1], n00)
        road_2 = (n00, nodes.index("%d%s0_1_" % (bi, block["id"]))) if block["id"] != "S" else None
        is_ramp = block["id"] in ("r", "R")
        if rng.rand() > PROHIBIT_SCENE_PROB:
            road = [road_1, road_2][int(rng.randint(0, 2))] if block["id"] != "C" else road_2
            road = road_1 if road is None else road
            on_left = bool(rng.rand() > 0.5) or (road is road_2 and is_ramp)
            ls = lanes_of(road)
            lane = ls[0] if on_left else ls[-1]
            L = desc["lanes"][lane]
            longitude = L["length"] - ACCIDENT_AREA_LEN
            accident_lanes += ls
            lat_num = int(desc["lane_width"] / CONE_LATERAL)
            longitude_num = int(ACCIDENT_AREA_LEN / CONE_LONGITUDE)
            lat_1 = [lat * CONE_LATERAL for lat in range(lat_num)]
            lat_2 = [lat_num * CONE_LATERAL] * (longitude_num + 1)
            lat_3 = [(lat_num - lat - 1) * CONE_LATERAL for lat in range(lat_num)]
            total = lat_num * 2 + longitude_num + 1
            left = 1 if on_left else -1
            for lg, lat in zip(range(-int(total / 2), int(total / 2)), lat_1 + lat_2 + lat_3):
                objs.append(dict(cls="cone", lane=lane, long=float(lg * CONE_LONGITUDE + longitude),
                                 lat=float(left * (lat - L["width"] / 2))))
        else:
            road = [road_1, road_2][int(rng.randint(0, 2))]
            road = road_1 if road is None else road
            on_left = bool(rng.rand() > 0.5) or (road is road_2 and is_ramp)
            ls = lanes_of(road)
            lane = ls[int(rng.randint(0, len(ls) - 1))] if on_left else ls[-1]
            L = desc["lanes"][lane]
            longitude = float(rng.rand() * L["length"] / 2 + L["length"] / 2)
            if rng.rand() > 0.5:  # break_down_scene: the vehicle type comes from the traffic manager's RNG
                objs.append(dict(cls="vehicle", lane=lane, long=longitude, lat=0.0, type_draw=n_type_draws))
                n_type_draws += 1
                objs.append(dict(cls="warning", lane=lane, long=longitude - ALERT_DIST, lat=0.0))
            else:
                objs.append(dict(cls="barrier", lane=lane, long=longitude, lat=0.0))
    return objs, accident_lanes, n_type_draws


def propose_respawn_traffic(desc, seed, density):
    """TrafficMode.Respawn: TrafficManager._create_respawn_vehicles / _create_vehicles_on_lane / _get_available_respawn_lanes
    (traffic_manager.py:188-222,236-239,292-309): one vehicle every 10 m on every lane of the map's respawn roads (a road
    listed by two blocks cancels out), driving from the first step on.  The density draw is commented out upstream, so
    every candidate spawns; vehicles leave for good when they run off the lanes (re-spawning is commented out too)."""
    rng = get_np_random(seed)
    if abs(density) < 1e-2:
        return []
    rl = mapdata.road_lookup(desc)
    roads = []
    for block in desc["blocks"]:
        for r in block["respawn_roads"]:
            r = tuple(r)
            if r in roads:
                roads.remove(r)
            else:
                roads.append(r)
    vehicles = []
    for r in roads:
        road = desc["roads"][rl[r]]
        for lid in range(road["first_lane"], road["first_lane"] + road["n_lanes"]):
            longs = [float(i * VEHICLE_GAP) for i in range(int(desc["lanes"][lid]["length"] / VEHICLE_GAP))]
            rng.shuffle(longs)
            for lg in longs:
                vtype = TYPE_KEYS[int(rng.choice(len(TYPE_KEYS), p=TRAFFIC_TYPE_PROB))]
                vehicles.append(dict(lane=lid, long=lg, vtype=vtype, policy_seed=int(rng.randint(0, MAX_RAND_INT))))
    return [dict(trigger_road=-1, vehicles=vehicles)]


def propose_traffic(desc, seed, density, skip_lanes=(), type_draws=0):
    """TrafficManager._create_vehicles_once (traffic_manager.py:239-290) -> list of groups
    [{trigger_road, vehicles:[{lane, long, vtype, policy_seed}]}] in *block* order, consuming the manager RNG exactly
    like the reference (shuffle, then per vehicle: type choice, policy seed)."""
    rng = get_np_random(seed)  # BaseManager seeds np_random with the global seed (base_manager.py:14)
    # broken-down vehicles of the object manager (reset before this manager) took their types from this stream
    pre_types = [TYPE_KEYS[int(rng.choice(len(TYPE_KEYS), p=TRAFFIC_TYPE_PROB))] for _ in range(type_draws)]
    rl = mapdata.road_lookup(desc)
    groups = []
    if abs(density) < 1e-2:
        return (groups, pre_types) if type_draws else groups
    for block in desc["blocks"][1:]:
        cands = []
        total_length = 0.0
        for lanes in block["spawn_lanes"]:
            for lid in lanes:
                l = desc["lanes"][lid]
                total_length += l["length"]
                if lid in skip_lanes:  # object_manager.accident_lanes (traffic_manager.py:256-257); the length still counts
                    continue
                for i in range(int(l["length"] / VEHICLE_GAP)):
                    cands.append((lid, float(i * VEHICLE_GAP)))
        total_spawn_points = int(math.floor(total_length / VEHICLE_GAP))
        total_vehicles = int(math.floor(total_spawn_points * density))
        order = list(range(len(cands)))
        rng.shuffle(order)
        selected = [cands[i] for i in order[:min(total_vehicles, len(cands))]]
        vehicles = []
        for lid, lg in selected:
            vtype = TYPE_KEYS[int(rng.choice(len(TYPE_KEYS), p=TRAFFIC_TYPE_PROB))]
            policy_seed = int(rng.randint(0, MAX_RAND_INT))
            vehicles.append(dict(lane=lid, long=lg, vtype=vtype, policy_seed=policy_seed))
        tr = block["trigger_road"]
        groups.append(dict(trigger_road=rl[(tr[0], tr[1])], vehicles=vehicles))
    return (groups, pre_types) if type_draws else groups


def build_scenario(desc, map_index, seed, num_agents=1, num_traffic=16, density=0.1, spawn_lane=None,
                   spawn_longitude=5.0, spawn_lateral=0.0, vehicle_model="default", agent_spawns=None,
                   traffic_mode="trigger", traffic_seed=None, auto_termination=False, accident_prob=0.0,
                   random_agent_model=False, idm_agent=False):
    """One scenario = V = num_agents + num_traffic spawn slots for map `desc` under global seed `seed`.
    idm_agent: the agents are driven by IDMPolicy (IDM_agent, agent_manager.py:79): their overtake timer starts at
    randint(0, LANE_CHANGE_FREQ) like every IDMPolicy's (idm_policy.py:185), drawn here from the vehicle's own seed (the
    reference takes the policy seed from the agent manager's generator: the stream is not reproduced, the range is)."""
    V = num_agents + num_traffic
    scen = np.zeros((), dtype=SCEN_DT)
    spawns = np.zeros(V, dtype=SPAWN_DT)
    spawns["lane"] = -1
    spawns["group"] = -1
    scen["map"] = map_index
    scen["trigger_road"][:] = -1
    scen["max_steps"] = 250 * len(desc["blocks"]) if auto_termination else 0  # base_env.py:318, map.num_blocks

    engine_rng = get_np_random(seed)  # BaseEngine.seed -> Randomizable.seed (base_engine.py:300-304)
    rl = mapdata.road_lookup(desc)

    # ---- traffic objects (object_manager.py:40-124; PRIORITY 9: reset before the agent and traffic managers, so its
    # spawn_object calls draw their engine seeds first).  They take the LAST traffic slots. ----
    objects, accident_lanes, n_type_draws = propose_objects(desc, seed, accident_prob) if num_traffic > 0 else ([], [], 0)
    obj_seeds = [int(engine_rng.randint(0, MAX_RAND_INT)) for _ in objects]

    # ---- agents (agent_manager.py:63-83: spawn_object -> engine.generate_seed()) ----
    if agent_spawns is None:
        if spawn_lane is None:
            r = desc["roads"][rl[(desc["nodes"].index(">"), desc["nodes"].index(">>"))]]
            spawn_lane = r["first_lane"] + 0
        agent_spawns = [dict(lane=spawn_lane, long=spawn_longitude, lat=spawn_lateral, dest=None)] * num_agents
    if random_agent_model:  # AgentManager._get_vehicles (agent_manager.py:63-73): ONE type draw per episode from the
        # agent manager's own stream (seeded with the episode seed), uniform over s / m / l / xl / default
        vehicle_model = TYPE_KEYS[int(get_np_random(seed).choice(len(TYPE_KEYS), p=[1 / len(TYPE_KEYS)] * len(TYPE_KEYS)))]
    for a in range(num_agents):
        sp = agent_spawns[a]
        obj_seed = int(engine_rng.randint(0, MAX_RAND_INT))
        params = sample_vehicle_params(vehicle_model, obj_seed)
        _fill_vehicle(spawns[a], desc, sp["lane"], sp["long"], sp["lat"], params)
        road = desc["roads"][desc["lanes"][sp["lane"]]["road"]]
        dest = sp.get("dest")
        if dest is None:
            dest = choose_destination(desc, seed, road["frm"], negative=road["negative"])
        _fill_route(spawns[a], desc, sp["lane"], dest)
        spawns[a]["group"] = -1
        spawns[a]["timer0"] = int(get_np_random(obj_seed).randint(0, 50)) if idm_agent else 0

    # ---- traffic (traffic_manager.py:239-290) ----
    # the manager RNG is re-seeded with the episode seed unless random_traffic (traffic_manager.py:348-350)
    tseed = seed if traffic_seed is None else traffic_seed
    if traffic_mode not in ("trigger", "hybrid", "respawn"):
        raise ValueError("No such mode named {}".format(traffic_mode))  # traffic_manager.py:68
    respawn = traffic_mode == "respawn"
    pre_types = []
    if num_traffic <= 0:
        groups = []
    elif respawn:
        groups = propose_respawn_traffic(desc, tseed, density)
    elif n_type_draws:
        groups, pre_types = propose_traffic(desc, tseed, density, skip_lanes=set(accident_lanes), type_draws=n_type_draws)
    else:
        groups = propose_traffic(desc, tseed, density, skip_lanes=set(accident_lanes))
    # objects first (they must not be dropped by the slot cap before ordinary traffic is)
    n_obj = min(len(objects), num_traffic)
    obj_dropped = len(objects) - n_obj
    for k in range(n_obj):
        o, r = objects[k], spawns[V - n_obj + k]
        lane = desc["lanes"][o["lane"]]
        road = desc["roads"][lane["road"]]
        if o["cls"] == "vehicle":  # break_down_scene: a vehicle of a traffic type that never drives
            params = sample_vehicle_params(pre_types[o["type_draw"]] if pre_types else "default", obj_seeds[k])
            _fill_vehicle(r, desc, o["lane"], o["long"], 0.0, params)
            r["kind"] = OBJ_VEHICLE
        else:
            x, y = mapdata.lane_position(lane, o["long"], o["lat"])
            h = mapdata.lane_heading_at(lane, o["long"])
            r["x"], r["y"], r["lane"] = x, y, o["lane"]
            r["wheelbase"], r["mass"], r["max_speed"], r["max_steer"], r["friction"] = 1.0, 1.0, 80.0, 0.1, 0.9
            if o["cls"] == "barrier":
                # BulletBoxShape((WIDTH/2, LENGTH/2, .)) with origin.setH(panda_heading(heading)): the reference passes the
                # heading in RADIANS where panda expects degrees (traffic_object.py:92), so the 2 m axis ends up at
                # heading * pi/180 - pi/2 in world coordinates -- reproduced
                r["kind"], r["length"], r["width"] = OBJ_BOX, BARRIER_LENGTH, BARRIER_WIDTH
                r["heading"] = h * math.pi / 180.0 - math.pi / 2
            else:
                rad = CONE_RADIUS if o["cls"] == "cone" else WARNING_RADIUS
                r["kind"], r["length"], r["width"], r["heading"] = OBJ_CYLINDER, 2 * rad, 2 * rad, h
        r["group"] = GROUP_NEVER
        r["n_ckpt"] = 2
        r["ckpt"][:] = -1
        r["ckpt_road"][:] = -1
        r["ckpt"][:2] = [road["frm"], road["to"]]
        r["ckpt_road"][0] = lane["road"]
        r["dest_lane"] = road["first_lane"] + road["n_lanes"] - 1
    V_traffic_end = V - n_obj
    slot = num_agents
    n_groups = 0
    dropped = 0
    for g in groups:
        if n_groups >= MAX_GROUPS:  # more traffic blocks than pgd_scenario.trigger_road holds: counted like slot-cap drops
            for v in g["vehicles"]:
                engine_rng.randint(0, MAX_RAND_INT)
                dropped += 1
            continue
        if not respawn:
            scen["trigger_road"][n_groups] = g["trigger_road"]
        for v in g["vehicles"]:
            obj_seed = int(engine_rng.randint(0, MAX_RAND_INT))  # consumed even if the slot cap drops the vehicle
            if slot >= V_traffic_end:
                dropped += 1
                continue
            params = sample_vehicle_params(v["vtype"], obj_seed)
            _fill_vehicle(spawns[slot], desc, v["lane"], v["long"], 0.0, params)
            road = desc["roads"][desc["lanes"][v["lane"]]["road"]]
            dest = choose_destination(desc, seed, road["frm"], negative=road["negative"])
            try:
                _fill_route(spawns[slot], desc, v["lane"], dest)
            except (KeyError, ValueError):
                # no route from this lane (e.g. dead-end ramp part): keep the lane's own road as route
                spawns[slot]["n_ckpt"] = 2
                spawns[slot]["ckpt"][:] = -1
                spawns[slot]["ckpt_road"][:] = -1
                spawns[slot]["ckpt"][:2] = [road["frm"], road["to"]]
                spawns[slot]["ckpt_road"][0] = desc["lanes"][v["lane"]]["road"]
                spawns[slot]["dest_lane"] = road["first_lane"] + road["n_lanes"] - 1
            spawns[slot]["group"] = -1 if respawn else n_groups  # respawn-mode vehicles drive from the first step
            spawns[slot]["timer0"] = int(get_np_random(v["policy_seed"]).randint(0, 50))  # idm_policy.py:185
            slot += 1
        n_groups += 0 if respawn else 1
    scen["n_groups"] = n_groups
    spawns["group"][V - n_obj:] = GROUP_NEVER  # (the loop above resets group to -1 only for unused slots)
    return scen, spawns, dict(dropped=dropped, n_traffic=slot - num_agents, n_objects=n_obj, objects_dropped=obj_dropped)


def resolve_lane_index(desc, lane_index):
    """(from node name, to node name, lane) as in vehicle_config.spawn_lane_index (pgdrive_env.py:77) -> map-local lane id."""
    frm, to, idx = lane_index
    nodes = desc["nodes"]
    if frm not in nodes or to not in nodes:
        raise KeyError("spawn_lane_index %r: no such node in map seed %s" % (lane_index, desc.get("seed")))
    rl = mapdata.road_lookup(desc)
    key = (nodes.index(frm), nodes.index(to))
    if key not in rl:
        raise KeyError("spawn_lane_index %r: no such road in map seed %s" % (lane_index, desc.get("seed")))
    road = desc["roads"][rl[key]]
    if not 0 <= idx < road["n_lanes"]:
        raise KeyError("spawn_lane_index %r: the road has %d lanes" % (lane_index, road["n_lanes"]))
    return road["first_lane"] + int(idx)


class ScenarioBank:
    def __init__(self, descs, seeds, num_agents=1, num_traffic=16, density=0.1, traffic_seeds=None, spawn_lane_index=None,
                 destination_node=None, **kw):
        """`spawn_lane_index` / `destination_node`: vehicle_config overrides of the spawn lane (default ('>', '>>', 0)) and of
        the destination (default: a seeded random socket of the last block), pgdrive_env.py:76-80, navigation.py:99-121."""
        scens, spawns, self.info = [], [], []
        for m, (d, s) in enumerate(zip(descs, seeds)):
            ts = None if traffic_seeds is None else int(traffic_seeds[m])
            if spawn_lane_index is not None or destination_node is not None:
                lane = resolve_lane_index(d, spawn_lane_index if spawn_lane_index is not None else (">", ">>", 0))
                if destination_node is not None and destination_node not in d["nodes"]:
                    raise KeyError("destination_node %r: no such node in map seed %s" % (destination_node, d.get("seed")))
                dest = None if destination_node is None else d["nodes"].index(destination_node)
                kw = dict(kw, agent_spawns=[dict(lane=lane, long=kw.get("spawn_longitude", 5.0),
                                                 lat=kw.get("spawn_lateral", 0.0), dest=dest)] * num_agents)
            sc, sp, info = build_scenario(d, m, s, num_agents, num_traffic, density, traffic_seed=ts, **kw)
            scens.append(sc)
            spawns.append(sp)
            self.info.append(info)
        self.scenarios = np.array(scens, dtype=SCEN_DT)
        self.spawns = np.concatenate(spawns)
        self.V = num_agents + num_traffic


# ----------------------------------------------------------------------------------------------------------------------
# multi-agent (envs/marl_envs/multi_agent_pgdrive.py, marl_inout_roundabout.py, manager/spawn_manager.py)
# ----------------------------------------------------------------------------------------------------------------------
RESPAWN_REGION_LONGITUDE = 8.0  # spawn_manager.py:27
RESPAWN_REGION_LATERAL = 3.0  # spawn_manager.py:28
MAX_VEHICLE_LENGTH, MAX_VEHICLE_WIDTH = 10.0, 2.5  # base_vehicle.py:83-84
ENTRANCE_LENGTH = 10  # first_block.py:21


def neg_road(desc, frm, to):
    """Road.__neg__ on node names (road.py:26-31)."""
    nodes = desc["nodes"]
    a, b = nodes[frm], nodes[to]
    if b.find("-") == -1:
        na, nb = "-" + b, "-" + a
    else:
        na, nb = b[b.find("-") + 1:], a[a.find("-") + 1:]
    return nodes.index(na), nodes.index(nb)


def roundabout_spawn_roads(desc):
    """MARoundaboutConfig.spawn_roads (marl_inout_roundabout.py:15-28): '>>'->'>>>' and the three exits, negated."""
    n = desc["nodes"]
    roads = [(n.index(">>"), n.index(">>>"))]
    for k in range(3):
        roads.append(neg_road(desc, n.index("1O%d_2_" % k), n.index("1O%d_3_" % k)))
    return roads


def intersection_spawn_roads(desc):
    """MAIntersectionConfig.spawn_roads (marl_intersection.py:14-20): '>>'->'>>>' and the three exits, negated."""
    n = desc["nodes"]
    roads = [(n.index(">>"), n.index(">>>"))]
    for k in range(3):
        roads.append(neg_road(desc, n.index("1X%d_0_" % k), n.index("1X%d_1_" % k)))
    return roads


def bottleneck_spawn_roads(desc):
    """MABottleneckConfig.spawn_roads (marl_bottleneck.py:12): '>>'->'>>>' and the far end of the Split block, negated."""
    n = desc["nodes"]
    return [(n.index(">>"), n.index(">>>")), neg_road(desc, n.index("2Y0_0_"), n.index("2Y0_1_"))]


def tollgate_spawn_roads(desc):
    """MATollConfig.spawn_roads (marl_tollgate.py:15): '>>'->'>>>' and the far end of the closing Merge block, negated."""
    n = desc["nodes"]
    return [(n.index(">>"), n.index(">>>")), neg_road(desc, n.index("3y0_0_"), n.index("3y0_1_"))]


def pg_spawn_roads(desc):
    """MULTI_AGENT_PGDRIVE_DEFAULT_CONFIG.spawn_roads (multi_agent_pgdrive.py:27): the straight of the first block."""
    n = desc["nodes"]
    return [(n.index(">>"), n.index(">>>"))]


MARL_SPAWN_ROADS = {"roundabout": roundabout_spawn_roads, "intersection": intersection_spawn_roads,
                    "bottleneck": bottleneck_spawn_roads, "tollgate": tollgate_spawn_roads, "pg": pg_spawn_roads}
# destination rule: the roundabout / intersection spawn managers draw a negated spawn road; the bottleneck env keeps the
# default SpawnManager.update_destination_for (spawn_manager.py:221-225), i.e. Navigation.update's own choice
MARL_AUTO_DEST = {"bottleneck", "tollgate", "pg"}
OBJ_BUILDING = 3  # PGD_OBJ_BUILDING


def spawn_slots(desc, spawn_roads):
    """SpawnManager._auto_fill_spawn_roads_randomly (spawn_manager.py:114-155): (road, lane, slot j) -> longitude;
    the j == 0 slots are the safe respawn places."""
    rl = mapdata.road_lookup(desc)
    exit_length = desc["exit_length"] - ENTRANCE_LENGTH
    num_slots = int(math.floor(exit_length / RESPAWN_REGION_LONGITUDE))
    slots, safe = [], []
    for (frm, to) in spawn_roads:
        road = desc["roads"][rl[(frm, to)]]
        for lane_idx in range(desc["lane_num"]):
            for j in range(num_slots):
                cfg = dict(lane=road["first_lane"] + lane_idx, long=0.5 * RESPAWN_REGION_LONGITUDE + j * RESPAWN_REGION_LONGITUDE,
                           lat=0.0, road=(frm, to))
                slots.append(cfg)
                if j == 0:
                    safe.append(cfg)
    return slots, safe


def map_buildings(desc):
    """Static buildings of the map (TollGate booths): dict(lane, x, y, heading, length, width) in block order."""
    return [b for blk in desc["blocks"] for b in blk.get("buildings", [])]


def resolve_spawn_roads(desc, spawn_roads):
    """`spawn_roads` of a multi-agent config as (from node, to node) pairs of this map: the reference passes `Road` objects
    (start_node / end_node); node-name pairs are accepted as well."""
    out = []
    for r in spawn_roads:
        a, b = (r.start_node, r.end_node) if hasattr(r, "start_node") else (r[0], r[1])
        if a not in desc["nodes"] or b not in desc["nodes"] or (desc["nodes"].index(a), desc["nodes"].index(b)) not in mapdata.road_lookup(desc):
            raise KeyError("spawn_roads: no road %r -> %r in the map" % (a, b))
        out.append((desc["nodes"].index(a), desc["nodes"].index(b)))
    if not out:
        raise ValueError("spawn_roads must not be empty (spawn_manager.py:107)")
    return out


def build_marl_scenario(desc, map_index, rng, num_agents, capacity=None, vehicle_model="default", kind="roundabout", fixed=None,
                        spawn_roads=None):
    """SpawnManager.reset (spawn_manager.py:68-101): `num_agents` of the spawn slots without replacement, jittered inside
    the slot, each with a random destination (RoundaboutSpawnManager.update_destination_for,
    marl_inout_roundabout.py:125-130); followed by the respawn table [safe place][destination].
    `rng` is a numpy RandomState (the reference leaves this manager unseeded).  `kind` selects the spawn roads; the
    destination rule is the same on the intersection map (InterectionSpawnManager, marl_intersection.py:57-62).
    `fixed`: {agent index: dict(spawn_lane_index, spawn_longitude, spawn_lateral, destination_node)} -- the agents named in
    `target_vehicle_configs` keep the placement they were given (`not_randomize`, multi_agent_pgdrive.py:96-107 and
    spawn_manager.py:58-69,91-100): the slot draw and the jitter still happen for every agent, then the given placement replaces
    the drawn one; a destination that was not given is still drawn."""
    spawn_roads = MARL_SPAWN_ROADS[kind](desc) if spawn_roads is None else resolve_spawn_roads(desc, spawn_roads)
    slots, safe = spawn_slots(desc, spawn_roads)
    infinite = num_agents == -1  # "as many vehicles as possible" (base_env.py:25): every spawn slot, in slot order
    if infinite:  # ... as far as the seats go (`max_agents`): the first `capacity` slots
        num_agents = min(len(slots), capacity or len(slots))
    A = capacity or num_agents
    if num_agents > len(slots) or num_agents > A:
        raise ValueError("Too many agents! We only accept %d agents" % min(len(slots), A))
    for a in (fixed or {}):
        if not 0 <= int(a) < num_agents:
            raise KeyError("target_vehicle_configs: 'agent%d' is not one of the %d initial agents (agent0 .. agent%d)" % (
                int(a), num_agents, num_agents - 1))
    auto = kind in MARL_AUTO_DEST

    def auto_dest(c):  # Navigation.update (navigation.py:99-121): last block's socket, first block's on a negative road
        road = desc["roads"][desc["lanes"][c["lane"]]["road"]]
        return choose_destination(desc, desc.get("seed", 0) if kind == "pg" else 0, road["frm"], negative=road["negative"])

    dests = [None] if auto else [neg_road(desc, *r)[1] for r in spawn_roads]  # end node of the negated spawn road
    P, Dn = len(safe), len(dests)
    buildings = map_buildings(desc)
    B = len(buildings)  # static bodies take the slots [A, A + B); the respawn table follows them
    recs = np.zeros(A + B + P * Dn, dtype=SPAWN_DT)
    recs["lane"] = -1
    recs["group"] = -1
    for k, b in enumerate(buildings):
        r = recs[A + k]
        lane = desc["lanes"][b["lane"]]
        road = desc["roads"][lane["road"]]
        r["x"], r["y"], r["lane"], r["kind"], r["group"] = b["x"], b["y"], b["lane"], OBJ_BUILDING, GROUP_NEVER
        # TollGateBuilding hands panda a heading in radians where degrees are expected (tollgate_building.py:18): the wall
        # ends up rotated by heading * pi / 180 -- reproduced
        r["heading"] = b["heading"] * math.pi / 180.0
        r["length"], r["width"] = b["length"], b["width"]
        r["wheelbase"], r["mass"], r["max_speed"], r["max_steer"], r["friction"] = 1.0, 1.0, 80.0, 0.1, 0.9
        r["n_ckpt"] = 2
        r["ckpt"][:] = -1
        r["ckpt_road"][:] = -1
        r["ckpt"][:2] = [road["frm"], road["to"]]
        r["ckpt_road"][0] = lane["road"]
        r["dest_lane"] = road["first_lane"] + road["n_lanes"] - 1
    # (spawn_manager.py:76-81: all slots in order when the number of agents is -1, else a draw without replacement)
    pick = np.arange(num_agents) if infinite else rng.choice(len(slots), num_agents, replace=False)
    lo, la = RESPAWN_REGION_LONGITUDE - MAX_VEHICLE_LENGTH, RESPAWN_REGION_LATERAL - MAX_VEHICLE_WIDTH
    for a, idx in enumerate(pick):
        c = slots[int(idx)]
        lon = c["long"] + rng.uniform(-lo / 2, lo / 2)
        lat = c["lat"] + rng.uniform(-la / 2, la / 2)
        params = sample_vehicle_params(vehicle_model, int(rng.randint(0, MAX_RAND_INT)))
        fx = (fixed or {}).get(a)
        if fx is not None:
            c = dict(c, lane=resolve_lane_index(desc, fx.get("spawn_lane_index") or (">", ">>", 0)))
            lon, lat = float(fx.get("spawn_longitude", 5.0)), float(fx.get("spawn_lateral", 0.0))
        _fill_vehicle(recs[a], desc, c["lane"], lon, lat, params)
        if fx is not None and fx.get("destination_node") is not None:
            if fx["destination_node"] not in desc["nodes"]:
                raise KeyError("destination_node %r: no such node in the map" % (fx["destination_node"], ))
            _fill_route(recs[a], desc, c["lane"], desc["nodes"].index(fx["destination_node"]))
        else:
            _fill_route(recs[a], desc, c["lane"], auto_dest(c) if auto else dests[int(rng.randint(0, Dn))])
    for p, c in enumerate(safe):
        for dn, dest in enumerate(dests):
            r = recs[A + B + p * Dn + dn]
            params = sample_vehicle_params(vehicle_model, int(rng.randint(0, MAX_RAND_INT)))
            _fill_vehicle(r, desc, c["lane"], c["long"], c["lat"], params)
            _fill_route(r, desc, c["lane"], auto_dest(c) if auto else dest)
    scen = np.zeros((), dtype=SCEN_DT)
    scen["map"] = map_index
    scen["trigger_road"][:] = -1
    return scen, recs, P, Dn, B


def parking_lot_roads(desc):
    """MAParkingLotConfig (marl_parking_lot.py:15-24,139-152): the three roads leading into the lot, the parking spaces
    driven outwards (spawn roads 1P{i}_5_ -> 1P{i}_6_) and inwards (destinations 1P{i}_1_ -> 1P{i}_2_)."""
    n = desc["nodes"]
    in_roads = [(n.index(">>"), n.index(">>>")), neg_road(desc, n.index("2T0_0_"), n.index("2T0_1_")),
                neg_road(desc, n.index("2T2_0_"), n.index("2T2_1_"))]
    k = 1
    out_roads, spaces = [], []
    while "1P%d_5_" % k in n:
        out_roads.append((n.index("1P%d_5_" % k), n.index("1P%d_6_" % k)))
        spaces.append((n.index("1P%d_1_" % k), n.index("1P%d_2_" % k)))
        k += 1
    return in_roads, out_roads, spaces


def build_parking_scenario(desc, map_index, rng, num_agents, capacity=None, vehicle_model="default"):
    """ParkingLotSpawnManager (marl_parking_lot.py:39-90) on top of SpawnManager.reset (spawn_manager.py:72-101): agents
    start on the three access roads or inside parking spaces; one that starts on a road is handed a free parking space as
    destination (distinct spaces), one that starts in a space drives out through a random access road.  The respawn table
    holds [access-road place][parking space]: upstream never re-fills the spaces themselves (its availability test
    compares the out-direction spawn road with the in-direction destination roads and so never succeeds)."""
    in_roads, out_roads, spaces = parking_lot_roads(desc)
    slots, safe = spawn_slots(desc, in_roads + out_roads)
    infinite = num_agents == -1
    if infinite:
        num_agents = len(slots)
    A = capacity or num_agents
    if num_agents > len(slots) or num_agents > A:
        raise ValueError("Too many agents! We only accept %d agents" % min(len(slots), A))
    S = len(spaces)
    places = [c for c in safe if c["road"] in in_roads]
    P, Dn = len(places), S
    recs = np.zeros(A + P * Dn, dtype=SPAWN_DT)
    recs["lane"] = -1
    recs["group"] = -1
    avail = list(range(S))
    pick = np.arange(num_agents) if infinite else rng.choice(len(slots), num_agents, replace=False)
    lo, la = RESPAWN_REGION_LONGITUDE - MAX_VEHICLE_LENGTH, RESPAWN_REGION_LATERAL - MAX_VEHICLE_WIDTH
    for a, idx in enumerate(pick):
        c = slots[int(idx)]
        lon = c["long"] + rng.uniform(-lo / 2, lo / 2)
        lat = c["lat"] + rng.uniform(-la / 2, la / 2)
        params = sample_vehicle_params(vehicle_model, int(rng.randint(0, MAX_RAND_INT)))
        _fill_vehicle(recs[a], desc, c["lane"], lon, lat, params)
        if c["road"] in in_roads:  # update_destination_for -> get_parking_space
            if not avail:
                raise ValueError("more agents on the access roads than parking spaces")
            k = avail.pop(int(rng.randint(0, len(avail))))
            _fill_route(recs[a], desc, c["lane"], spaces[k][1])
            recs[a]["aux"] = k + 1
        else:
            road = in_roads[int(rng.randint(0, len(in_roads)))]
            _fill_route(recs[a], desc, c["lane"], neg_road(desc, *road)[1])
    for p, c in enumerate(places):
        for k in range(S):
            r = recs[A + p * Dn + k]
            params = sample_vehicle_params(vehicle_model, int(rng.randint(0, MAX_RAND_INT)))
            _fill_vehicle(r, desc, c["lane"], c["long"], c["lat"], params)
            _fill_route(r, desc, c["lane"], spaces[k][1])
            r["aux"] = k + 1
    scen = np.zeros((), dtype=SCEN_DT)
    scen["map"] = map_index
    scen["trigger_road"][:] = -1
    scen["aux"] = sum(1 << k for k in avail)
    return scen, recs, P, Dn, 0


class MarlScenarioBank:
    """`n_variants` random initial placements over one multi-agent map (scenarios differ only in spawn choice)."""
    def __init__(self, desc, num_agents, capacity=None, n_variants=16, seed=0, kind="roundabout", fixed=None, spawn_roads=None):
        """`desc`: one map description, or a list of them (kind "pg": the generic multi-agent env over several generated
        maps, `n_variants` placements per map; the respawn table has the same shape on every map)."""
        rng = np.random.RandomState(seed)
        scens, recs = [], []
        descs = desc if isinstance(desc, (list, tuple)) else [desc]
        shape = None
        for m, dm in enumerate(descs):
            for _ in range(n_variants):
                if kind == "parking":
                    sc, rc, self.P, self.Dn, self.B = build_parking_scenario(dm, m, rng, num_agents, capacity)
                else:
                    sc, rc, self.P, self.Dn, self.B = build_marl_scenario(dm, m, rng, num_agents, capacity, kind=kind, fixed=fixed,
                                                                          spawn_roads=spawn_roads)
                if shape is None:
                    shape = (self.P, self.Dn, self.B)
                elif shape != (self.P, self.Dn, self.B):
                    raise ValueError("maps of one multi-agent bank need equal respawn tables: %r vs %r" % (
                        shape, (self.P, self.Dn, self.B)))
                scens.append(sc)
                recs.append(rc)
        self.scenarios = np.array(scens, dtype=SCEN_DT)
        self.spawns = np.concatenate(recs)
        self.infinite = num_agents == -1  # AgentManager.allow_respawn ignores the agent count then (agent_manager.py:316-323)
        if self.infinite:  # every spawn slot is filled at the start: the records that hold a lane
            num_agents = int((recs[0]["lane"][:capacity or len(recs[0])] >= 0).sum()) if capacity else \
                len(recs[0]) - self.B - self.P * self.Dn
        self.A = capacity or num_agents
        self.V = self.A + self.B  # agents + static bodies (toll booths)
        self.num_agents = num_agents
        self.stride = self.V + self.P * self.Dn
        self.info = []
