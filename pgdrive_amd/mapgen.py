"""Host-side procedural map generator: our own block-incremental generation (BIG) of PGDrive maps.

Runs once per map seed on the host and yields the flat *map description* consumed by `mapdata.MapBank` (lanes, roads,
block metadata) — no dependency on the reference at run time.

This module is NOT a re-design and does not claim to be one: the maps must come out bit-identical to the reference's (same
topology, same coordinates, same RNG draw order), so every block class below follows the reference's construction step by
step -- the same tool lanes, the same sequence of `rng` draws, the same radii and magic numbers -- on its own data model
(tuple roads, `SLane` / `CLane`, `Net`).  The similarity to `component/blocks/*.py` is by necessity; what is ours is the
flat description format, the host / device split (generation once on the host, immutable tables on the GPU) and the checks
against the reference's exports.  It restates, in float64 like the reference:

* BIG search (forward / destruct / sibling / back, MAX_TRIAL = 2)      component/algorithm/BIG.py:27-151
* block type distribution V2                                            component/algorithm/blocks_prob_dist.py:31-49
* parameter spaces + seeded sampling                                    utils/space.py:152-306, base_runnable.py:81-88
* lane construction helpers (bend + straight, side lanes, adverse road) component/blocks/create_block_utils.py:16-230
* overlap test on 1 m samples                                           utils/scene_utils.py:40-135
* blocks: first block, Straight, Curve, In/Out ramp, (T-)intersection, Roundabout
                                                                         component/blocks/{first_block,straight,curve,ramp,
                                                                         intersection,t_intersection,roundabout}.py
* socket / respawn-road / spawn-lane bookkeeping                         component/blocks/pg_block.py:48-213

Checked lane-for-lane against descriptions exported from the reference's own BIG: the 100 PGDrive-v0 seeds plus 14
extra cases (other block counts / lane counts / widths, explicit sequences through every block type) — identical
topology, line types and block bookkeeping, coordinates bit-identical (tests/test_mapgen.py).  Parameters are
float32-valued gym Box samples turned into python floats (utils/config.py:207-216); all arithmetic is float64.
"""
import copy
import math
from collections import OrderedDict, deque

import numpy as np

from .scenario import get_np_random

NONE, BROKEN, CONTINUOUS, SIDE = 0, 1, 2, 3  # LineType codes of mapdata (constants.py:203-217)
GREY, YELLOW = 0, 1
DEC_START, DEC_END = "decoration", "decoration_"  # constants.py:76-81
SIDEWALK_WIDTH, SIDEWALK_LINE_DIST = 3.0, 0.6  # constants.py:239-242


# ----------------------------------------------------------------------------------------------------------------------
# lanes (component/lane/straight_lane.py, circular_lane.py)
# ----------------------------------------------------------------------------------------------------------------------
def wrap_to_pi(x):
    return ((x + np.pi) % (2 * np.pi)) - np.pi


class SLane:
    kind = 0

    def __init__(self, start, end, width=4.0, line_types=(BROKEN, BROKEN), forbidden=False, speed_limit=1000, priority=0):
        self.start = np.array(start, dtype=np.float64)
        self.end = np.array(end, dtype=np.float64)
        self.width = width
        self.line_types = list(line_types) if line_types else [BROKEN, BROKEN]
        self.line_color = [GREY, GREY]
        self.forbidden, self.speed_limit, self.priority = forbidden, speed_limit, priority
        self.update_properties()

    def update_properties(self):
        d = self.end - self.start
        self.length = math.sqrt(d[0]**2 + d[1]**2)
        self.heading = math.atan2(d[1], d[0])
        self.direction = d / self.length
        self.direction_lateral = np.array([-self.direction[1], self.direction[0]])

    def position(self, lon, lat):
        return self.start + lon * self.direction + lat * self.direction_lateral

    def heading_at(self, lon):
        return self.heading

    def local_coordinates(self, p):
        dx, dy = p[0] - self.start[0], p[1] - self.start[1]
        return (float(dx * self.direction[0] + dy * self.direction[1]),
                float(dx * self.direction_lateral[0] + dy * self.direction_lateral[1]))


class CLane:
    kind = 1

    def __init__(self, center, radius, start_phase, end_phase, clockwise=True, width=4.0, line_types=(BROKEN, BROKEN),
                 forbidden=False, speed_limit=1000, priority=0):
        self.center = np.array(center, dtype=np.float64)
        self.radius = radius
        self.start_phase, self.end_phase = start_phase, end_phase
        self.direction = 1 if clockwise else -1
        self.width = width
        self.line_types = list(line_types) if line_types is not None else None
        self.line_color = [GREY, GREY]
        self.forbidden, self.speed_limit, self.priority = forbidden, speed_limit, priority
        self.update_properties()

    def update_properties(self):
        self.length = self.radius * (self.end_phase - self.start_phase) * self.direction
        self.start = self.position(0, 0)
        self.end = self.position(self.length, 0)

    def position(self, lon, lat):
        phi = self.direction * lon / self.radius + self.start_phase
        r = self.radius - lat * self.direction
        return self.center + r * np.array([math.cos(phi), math.sin(phi)])

    def heading_at(self, lon):
        phi = self.direction * lon / self.radius + self.start_phase
        return phi + math.pi / 2 * self.direction

    def local_coordinates(self, p):
        dx, dy = p[0] - self.center[0], p[1] - self.center[1]
        phi = math.atan2(dy, dx)
        phi = self.start_phase + wrap_to_pi(phi - self.start_phase)
        r = math.sqrt(dx**2 + dy**2)
        return self.direction * (phi - self.start_phase) * self.radius, self.direction * (self.radius - r)


# ----------------------------------------------------------------------------------------------------------------------
# roads and road network (component/road/road.py, road_network.py)
# ----------------------------------------------------------------------------------------------------------------------
def neg(road):
    """Road.__neg__ (road.py:26-31)"""
    a, b = road
    k = b.find("-")
    if k == -1:
        return ("-" + b, "-" + a)
    return (b[k + 1:], a[k + 1:])


def is_negative(road):
    return road[1].find("-") != -1


def is_valid(road):
    return not (road[0] == DEC_START and road[1] == DEC_END)


class Net:
    def __init__(self):
        self.graph = {}

    def add_lane(self, a, b, lane):
        self.graph.setdefault(a, {}).setdefault(b, []).append(lane)

    def lanes(self, road):
        return self.graph[road[0]][road[1]]

    def decoration_lanes(self):
        return self.graph[DEC_START][DEC_END] if DEC_START in self.graph else []

    def add(self, other):
        """RoadNetwork.add (road_network.py:36-48): inner dicts are shared with the block's own network."""
        s1 = set(self.graph) - {DEC_START, DEC_END}
        s2 = set(other.graph) - {DEC_START, DEC_END}
        if s1 & s2:
            raise ValueError("Same start node {} in two road network".format(s1 & s2))
        dec = self.decoration_lanes() + other.decoration_lanes()
        self.graph.update(copy.copy(other.graph))
        if dec:
            self.graph.pop(DEC_START, None)
            self.graph[DEC_START] = {DEC_END: dec}

    def remove(self, other):
        """RoadNetwork.__isub__ (road_network.py:50-59)"""
        for k in self.graph.keys() & (other.graph.keys() - {DEC_START, DEC_END}):
            self.graph.pop(k, None)
        if DEC_START in other.graph:
            for lane in other.graph[DEC_START][DEC_END]:
                if lane in self.graph[DEC_START][DEC_END]:
                    self.graph[DEC_START][DEC_END].remove(lane)

    def positive_lanes(self):
        out = []
        for a, td in self.graph.items():
            for b, ls in td.items():
                if not is_negative((a, b)) and is_valid((a, b)):
                    out.append(ls)
        return out

    def bfs_paths(self, start, goal):
        queue = [(start, [start])]
        while queue:
            node, path = queue.pop(0)
            if node not in self.graph:
                yield []
                continue
            for nxt in [n for n in self.graph[node].keys() if n not in path]:
                if nxt == goal:
                    yield path + [nxt]
                elif nxt in self.graph:
                    queue.append((nxt, path + [nxt]))

    def remove_all_roads(self, start, end):
        ret = []
        for path in list(self.bfs_paths(start, end)):
            for i, node in enumerate(path[:-1], 1):
                if node in self.graph and path[i] in self.graph[node]:
                    ret += self.graph[node].pop(path[i])
                    if len(self.graph[node]) == 0:
                        self.graph.pop(node)
        return ret


# ----------------------------------------------------------------------------------------------------------------------
# overlap test (utils/scene_utils.py:40-135)
# ----------------------------------------------------------------------------------------------------------------------
def _straight_contour(lanes, extra):
    pts = []
    for lane, d in ((lanes[0], -1), (lanes[-1], 1)):
        pts.append(lane.position(0.1, d * (lane.width / 2.0 + extra)))
        pts.append(lane.position(lane.length - 0.1, d * (lane.width / 2.0 + extra)))
    return pts


def _curve_contour(lanes, extra):
    pts = []
    for lane, ld in ((lanes[0], -1), (lanes[-1], 1)):
        pi_2 = np.pi / 2.0
        pts += [lane.position(0.1, ld * (lane.width / 2.0 + extra)), lane.position(lane.length - 0.1, ld * (lane.width / 2.0 + extra))]
        start_phase = (lane.start_phase // pi_2) * pi_2
        start_phase += pi_2 if lane.direction == 1 else 0
        for k in range(4):
            phi = start_phase + k * pi_2 * lane.direction
            if lane.direction * phi > lane.direction * lane.end_phase:
                break
            pts.append(lane.center + (lane.radius - ld * (lane.width / 2.0 + extra) * lane.direction) *
                       np.array([math.cos(phi), math.sin(phi)]))
    return pts


def road_bbox(lanes, extra=3):
    pts = np.array(_curve_contour(lanes, extra) if lanes[0].kind == 1 else _straight_contour(lanes, extra))
    return pts[:, 0].max(), pts[:, 0].min(), pts[:, 1].max(), pts[:, 1].min()


def check_lane_on_road(net, lane, positive=0, ignored=None, skip=False):
    """True when `lane` (sampled every metre at lateral offset positive*width/2) lies on a lane of `net`."""
    if skip:
        return True
    x_max_2, x_min_2, y_max_2, y_min_2 = road_bbox([lane])
    samples = None
    for a, td in net.graph.items():
        for b, lanes in td.items():
            if ignored and (a, b) == ignored:
                continue
            if (a, b) == (DEC_START, DEC_END) or len(lanes) == 0:
                continue
            x_max_1, x_min_1, y_max_1, y_min_1 = road_bbox(lanes)
            if x_min_1 > x_max_2 or x_min_2 > x_max_1 or y_min_1 > y_max_2 or y_min_2 > y_max_1:
                continue
            if samples is None:
                samples = [lane.position(i, positive * lane.width / 2.0) for i in range(1, int(lane.length), 1)]
            for l in lanes:
                for p in samples:
                    lon, lat = l.local_coordinates(p)
                    if math.fabs(lat) <= l.width / 2.0 and 0 <= lon <= l.length:
                        return True
    return False


# ----------------------------------------------------------------------------------------------------------------------
# construction helpers (component/blocks/create_block_utils.py)
# ----------------------------------------------------------------------------------------------------------------------
def vertical_vectors(v):
    length = math.sqrt(v[0]**2 + v[1]**2)
    return (v[1] / length, -v[0] / length), (-v[1] / length, v[0] / length)


def create_bend_straight(prev, following_len, radius, angle, clockwise=True, width=4.0, line_types=None, forbidden=False,
                         speed_limit=20, priority=0):
    bd = 1 if clockwise else -1
    center = prev.position(prev.length, bd * radius)
    x, y = prev.direction_lateral
    start_phase = 0
    if y == 0:
        start_phase = 0 if x < 0 else -np.pi
    elif x == 0:
        start_phase = np.pi / 2 if y < 0 else -np.pi / 2
    else:
        base = np.arctan(y / x)
        if x < 0:
            start_phase = base
        elif y < 0:
            start_phase = np.pi + base
        elif y > 0:
            start_phase = -np.pi + base
    end_phase = start_phase + angle
    if not clockwise:
        start_phase = start_phase - np.pi
        end_phase = start_phase - angle
    bend = CLane(center, radius, start_phase, end_phase, clockwise, width, line_types, forbidden, speed_limit, priority)
    length = 2 * radius * angle / 2
    bend_end = bend.position(length, 0)
    vv = vertical_vectors(bend_end - center)
    nxt = np.asarray(vv[0] if not clockwise else vv[1])
    straight = SLane(bend_end, nxt * following_len + bend_end, width, line_types, forbidden, speed_limit, priority)
    return bend, straight


def create_wave_lanes(pre_lane, lateral_dist, wave_length, last_straight_length, lane_width, toward_left=True):
    """create_wave_lanes (create_block_utils.py:308-329): two opposite arcs that shift a lane sideways by `lateral_dist`
    over `wave_length`, followed by a straight lane."""
    angle = np.pi - 2 * np.arctan(wave_length / (2 * lateral_dist))
    radius = wave_length / (2 * math.sin(angle))
    c1, pre = create_bend_straight(pre_lane, 10, radius, angle, False if toward_left else True, lane_width, [NONE, NONE])
    s, e = pre.position(-10, 0), pre.position(pre.length - 10, 0)
    pre.start, pre.end = np.array(s, dtype=np.float64), np.array(e, dtype=np.float64)  # reset_start_end
    pre.update_properties()
    c2, straight = create_bend_straight(pre, last_straight_length, radius, angle, True if toward_left else False,
                                        lane_width, [NONE, NONE])
    return c1, c2, straight


def extend_straight(lane, extend_length, line_types):
    new = copy.deepcopy(lane)
    new.start = lane.end
    new.end = lane.position(lane.length + extend_length, 0)
    new.line_types = list(line_types)
    new.update_properties()
    return new


def create_road_from(lane, lane_num, road, net_add, net_check, toward_smaller=True, ignore=(None, None),
                     center_line_type=CONTINUOUS, detect_one_side=True, side_type=SIDE, inner_type=BROKEN,
                     center_color=YELLOW, skip=False):
    """CreateRoadFrom (create_block_utils.py:60-148): `lane` is the outermost lane, lane_num-1 siblings are derived."""
    lane_num -= 1
    origin = lane
    lanes = []
    w = lane.width
    for i in range(lane_num, 0, -1):
        side = copy.deepcopy(lane)
        if lane.kind == 0:
            off = -w if toward_smaller else w
            s, e = side.position(0, off), side.position(side.length, off)
            side.start, side.end = s, e
        else:
            cw = lane.direction == 1
            if not toward_smaller:
                side.radius = lane.radius - w if cw else lane.radius + w
            else:
                side.radius = lane.radius + w if cw else lane.radius - w
            side.update_properties()
        if i == 1:
            side.line_types = [center_line_type, inner_type] if toward_smaller else [inner_type, side_type]
        else:
            side.line_types = [inner_type, inner_type]
        lanes.append(side)
        lane = side
    if toward_smaller:
        lanes.reverse()
        lanes.append(origin)
        origin.line_types = [inner_type if len(lanes) > 1 else center_line_type, side_type]
    else:
        lanes.insert(0, origin)
        if len(lanes) > 1:
            origin.line_types = (origin.line_types[0], lanes[-1].line_types[0])
    factor = (SIDEWALK_WIDTH + SIDEWALK_LINE_DIST + w / 2.0) * 2.0 / w
    if not detect_one_side:
        no_cross = not (check_lane_on_road(net_check, origin, factor, ignore, skip) or
                        check_lane_on_road(net_check, lanes[0], -0.95, ignore, skip))
    else:
        no_cross = not check_lane_on_road(net_check, origin, factor, ignore, skip)
    for l in lanes:
        net_add.add_lane(road[0], road[1], l)
    if lane_num == 0:
        lanes[-1].line_types = [center_line_type, side_type]
    lanes[0].line_color = [center_color, GREY]
    return no_cross


def create_adverse_road(positive_road, net_get, net_check, ignore=(None, None), center_line_type=CONTINUOUS,
                        side_type=SIDE, inner_type=BROKEN, center_color=YELLOW, skip=False):
    adverse = neg(positive_road)
    lanes = net_get.lanes(positive_road)
    ref = lanes[-1]
    num = len(lanes) * 2
    w = ref.width
    if ref.kind == 0:
        sym = SLane(ref.position(lanes[-1].length, -(num - 1) * w), ref.position(0, -(num - 1) * w), w, lanes[-1].line_types,
                    ref.forbidden, ref.speed_limit, ref.priority)
    else:
        cw = not (ref.direction == 1)
        radius = ref.radius + (num - 1) * w if not cw else ref.radius - (num - 1) * w
        sym = CLane(ref.center, radius, ref.end_phase, ref.start_phase, cw, w, ref.line_types, ref.forbidden,
                    ref.speed_limit, ref.priority)
    ok = create_road_from(sym, int(num / 2), adverse, net_get, net_check, ignore=ignore, side_type=side_type,
                          inner_type=inner_type, center_line_type=center_line_type, center_color=center_color, skip=skip)
    net_get.lanes(positive_road)[0].line_color = [center_color, GREY]
    return ok


def create_two_way_road(road_to_change, net_get, net_check, new_road_name=None, center_line_type=CONTINUOUS, side_type=SIDE,
                        inner_type=BROKEN, skip=False):
    """CreateTwoWayRoad (create_block_utils.py:233-295): the same strip of asphalt in the opposite direction under a new
    road name, so that it can be driven both ways (parking spaces)."""
    adverse = (road_to_change[1], road_to_change[0]) if new_road_name is None else new_road_name
    lanes = net_get.lanes(road_to_change)
    ref = lanes[-1]
    num = len(lanes)
    w = ref.width
    if ref.kind == 0:
        sym = SLane(ref.position(lanes[-1].length, -(num - 1) * w), ref.position(0, -(num - 1) * w), w, lanes[-1].line_types,
                    ref.forbidden, ref.speed_limit, ref.priority)
    else:
        cw = not (ref.direction == 1)
        radius = ref.radius + (num - 1) * w if not cw else ref.radius - (num - 1) * w
        sym = CLane(ref.center, radius, ref.end_phase, ref.start_phase, cw, w, ref.line_types, ref.forbidden,
                    ref.speed_limit, ref.priority)
    return create_road_from(sym, num, adverse, net_get, net_check, side_type=side_type, inner_type=inner_type,
                            center_line_type=center_line_type, skip=skip)


# ----------------------------------------------------------------------------------------------------------------------
# parameter spaces (utils/space.py:258-306): name -> ("box", min, max) | ("disc", min, max) | ("const", v)
# ----------------------------------------------------------------------------------------------------------------------
SPACES = {
    "S": {"length": ("box", 40.0, 80.0)},
    "C": {"length": ("box", 40.0, 80.0), "radius": ("box", 25.0, 60.0), "angle": ("box", 45, 135), "dir": ("disc", 0, 1)},
    "X": {"radius": ("const", 10), "change_lane_num": ("disc", 0, 1), "decrease_increase": ("disc", 0, 1)},
    "O": {"exit_radius": ("box", 5, 15), "inner_radius": ("box", 15, 45), "angle": ("const", 60)},
    "T": {"radius": ("const", 10), "t_type": ("disc", 0, 2), "change_lane_num": ("disc", 0, 1),
          "decrease_increase": ("disc", 0, 1)},
    "r": {"length": ("box", 20, 40)},
    "R": {"length": ("box", 20, 40)},
    "I": {},
    "y": {"length": ("box", 20, 50), "lane_num": ("disc", 1, 2), "bottle_len": ("const", 20)},  # BOTTLENECK_PARAMETER
    "Y": {"length": ("box", 20, 50), "lane_num": ("disc", 1, 2), "bottle_len": ("const", 20)},
    "$": {"length": ("box", 20, 50), "lane_num": ("disc", 1, 2), "bottle_len": ("const", 20)},  # TollGate reuses it
    "P": {"one_side_vehicle_number": ("disc", 2, 10), "radius": ("const", 4), "length": ("const", 8)},  # PARKING_LOT_PARAMETER
}


def sample_space(space, seed):
    """ParameterSpace.seed(seed) + sample(): every Box owns an RNG seeded with the same seed (space.py:100-104,
    430-457); values are float32 (or floored int64) and become python scalars in Config._set_item (config.py:207-216)."""
    out = {}
    for k, spec in space.items():
        rng = get_np_random(seed)
        if spec[0] == "box":
            lo, hi = np.float32(spec[1]), np.float32(spec[2])
            out[k] = float(np.float32(rng.uniform(low=lo, high=hi)))
        elif spec[0] == "disc":
            lo, hi = np.int64(spec[1]), np.int64(spec[2]) + 1
            out[k] = int(np.floor(rng.uniform(low=lo, high=hi)))
        else:
            v = np.float32(spec[1])
            out[k] = float(np.float32(rng.uniform(low=v, high=v)))
    return out


# ----------------------------------------------------------------------------------------------------------------------
# blocks (component/blocks/*.py)
# ----------------------------------------------------------------------------------------------------------------------
class Socket:
    def __init__(self, pos, negr=None):
        self.pos, self.neg = pos, negr
        self.index = None

    def set_index(self, name, i):
        self.index = "{}-socket{}".format(name, i)


def real_index(name, i):
    return "{}-socket{}".format(name, i)


class Block:
    ID = "B"
    RADIUS, ANGLE_RAMP, CONNECT_PART_LEN, RAMP_LEN = 40, 10, 20, 15  # Ramp (ramp.py:29-35)

    def __init__(self, index, pre_socket, gnet, seed, skip_check=False):
        self.index, self.name = index, str(index) + self.ID
        self.gnet, self.net = gnet, Net()
        self.skip = skip_check
        self.rng = get_np_random(seed)
        self.respawn_roads = []
        self.sockets = OrderedDict()
        self.trials = 0
        self.part_idx = self.road_idx = 0
        self.pre_socket = pre_socket
        self.pre_socket_index = pre_socket.index if pre_socket is not None else None
        self.config = {}
        self.sample_parameters()  # BaseRunnable.__init__ draws once (base_runnable.py:31)
        if index != 0:
            self.pos_lanes = gnet.lanes(pre_socket.pos)
            self.neg_lanes = gnet.lanes(pre_socket.neg)
            self.pos_lane_num = len(self.pos_lanes)
            self.pos_basic = self.pos_lanes[-1]
            self.lane_width = self.pos_basic.width

    def sample_parameters(self):
        self.config.update(sample_space(SPACES[self.ID], int(self.rng.randint(low=0, high=int(1e6)))))

    # -- naming (pg_block.py:183-201)
    def node(self, part, road):
        return str(self.index) + self.ID + str(part) + "_" + str(road) + "_"

    def set_part(self, x):
        self.part_idx, self.road_idx = x, 0

    def add_node(self):
        self.road_idx += 1
        return self.node(self.part_idx, self.road_idx - 1)

    def add_socket(self, s):
        if s.index is None:
            s.set_index(self.name, len(self.sockets))
        self.sockets[s.index] = s

    def socket_from_positive(self, road):
        return Socket(road, neg(road))

    def get_socket(self, i):
        key = list(self.sockets)[i] if isinstance(i, (int, np.integer)) else i
        return self.sockets[key]

    def socket_indices(self):
        return list(self.sockets.keys())

    # -- construct / destruct (base_block.py:72-110)
    def construct(self, extra_config=None):
        self.sample_parameters()
        if extra_config:
            self.config.update(extra_config)
        self.clear()
        self.trials += 1
        ok = self.build()
        self.gnet.add(self.net)
        return ok

    def clear(self):
        self.gnet.remove(self.net)
        self.net.graph.clear()
        self.part_idx = self.road_idx = 0
        self.respawn_roads = []
        self.sockets.clear()

    def destruct(self):
        self.clear()

    def respawn_lanes(self):
        return [self.net.lanes(r) for r in self.respawn_roads]

    def intermediate_spawn_lanes(self):
        trig = self.net.positive_lanes()
        for ls in self.respawn_lanes():
            if ls not in trig:
                trig.append(ls)
        return trig

    def rf(self, lane, n, road, **kw):
        return create_road_from(lane, n, road, self.net, self.gnet, skip=self.skip, **kw)

    def ar(self, road, **kw):
        return create_adverse_road(road, self.net, self.gnet, skip=self.skip, **kw)

    def on_road(self, lane, positive):
        return check_lane_on_road(self.gnet, lane, positive, skip=self.skip)


class FirstBlock(Block):
    """first_block.py:12-89"""
    ID = "I"

    def __init__(self, gnet, lane_width, lane_num, length=50):
        super().__init__(0, None, gnet, 0)
        basic = SLane([0, lane_width * (lane_num - 1)], [10, lane_width * (lane_num - 1)], lane_width, (BROKEN, SIDE))
        r1 = (">", ">>")
        self.rf(basic, lane_num, r1)
        self.ar(r1)
        nxt = extend_straight(basic, length - 10, [BROKEN, SIDE])
        r2 = (">>", ">>>")
        self.rf(nxt, lane_num, r2)
        self.ar(r2)
        gnet.add(self.net)
        s = self.socket_from_positive(r2)
        s.set_index(self.name, 0)
        self.add_socket(s)
        self.respawn_roads = [r2]


class Straight(Block):
    ID = "S"

    def build(self):
        self.set_part(0)
        new_lane = extend_straight(self.pos_basic, self.config["length"], [BROKEN, SIDE])
        road = (self.pre_socket.pos[1], self.add_node())
        ok = self.rf(new_lane, self.pos_lane_num, road)
        ok = self.ar(road) and ok
        self.add_socket(Socket(road, neg(road)))
        return ok


class Curve(Block):
    ID = "C"

    def build(self):
        p = self.config
        road = (self.pre_socket.pos[1], self.add_node())
        curve, straight = create_bend_straight(self.pos_basic, p["length"], p["radius"], np.deg2rad(p["angle"]), p["dir"],
                                               self.pos_basic.width, (BROKEN, SIDE))
        ok = self.rf(curve, self.pos_lane_num, road)
        ok = self.ar(road) and ok
        road = (road[1], self.add_node())
        ok = self.rf(straight, self.pos_lane_num, road) and ok
        ok = self.ar(road) and ok
        self.add_socket(self.socket_from_positive(road))
        return ok


class InRamp(Block):
    ID = "r"
    EXTRA_PART, SOCKET_LEN = 10, 20

    def build(self):
        acc_len = self.config["length"]
        ok = True
        self.set_part(0)
        sa, ca = math.sin(np.deg2rad(self.ANGLE_RAMP)), math.cos(np.deg2rad(self.ANGLE_RAMP))
        longitude = sa * self.RADIUS * 2 + ca * self.CONNECT_PART_LEN + self.RAMP_LEN
        n = self.pos_lane_num
        extend_lane = extend_straight(self.pos_basic, longitude + self.EXTRA_PART, [BROKEN, CONTINUOUS])
        extend_road = (self.pre_socket.pos[1], self.add_node())
        ok = self.rf(extend_lane, n, extend_road, side_type=CONTINUOUS) and ok
        self.net.lanes(extend_road)[-1].line_types = [BROKEN if n != 1 else CONTINUOUS, CONTINUOUS]
        ok = self.ar(extend_road) and ok
        self.net.lanes(neg(extend_road))[-1].line_types = [NONE if n == 1 else BROKEN, SIDE]
        acc_side = extend_straight(extend_lane, acc_len + self.lane_width, [extend_lane.line_types[0], SIDE])
        acc_road = (extend_road[1], self.add_node())
        ok = self.rf(acc_side, n, acc_road, side_type=CONTINUOUS) and ok
        ok = self.ar(acc_road) and ok
        self.net.lanes(acc_road)[-1].line_types = [CONTINUOUS if n == 1 else BROKEN, BROKEN]
        socket_side = extend_straight(acc_side, self.SOCKET_LEN, acc_side.line_types)
        socket_road = (acc_road[1], self.add_node())
        ok = self.rf(socket_side, n, socket_road, side_type=CONTINUOUS) and ok
        ok = self.ar(socket_road) and ok
        self.add_socket(self.socket_from_positive(socket_road))
        # ramp part
        self.set_part(1)
        lateral = (1 - ca) * self.RADIUS * 2 + sa * self.CONNECT_PART_LEN
        end = extend_lane.position(self.EXTRA_PART + self.RAMP_LEN, lateral + self.lane_width)
        start = extend_lane.position(self.EXTRA_PART, lateral + self.lane_width)
        LT = (CONTINUOUS, CONTINUOUS)
        straight_part = SLane(start, end, self.lane_width, LT, speed_limit=12)
        straight_road = (self.add_node(), self.add_node())
        self.net.add_lane(straight_road[0], straight_road[1], straight_part)
        ok = (not self.on_road(straight_part, 0.95)) and ok
        self.respawn_roads.append(straight_road)
        bend_1, connect = create_bend_straight(straight_part, self.CONNECT_PART_LEN, self.RADIUS, np.deg2rad(self.ANGLE_RAMP),
                                               False, self.lane_width, LT, speed_limit=12)
        bend_1_road = (straight_road[1], self.add_node())
        connect_road = (bend_1_road[1], self.add_node())
        self.net.add_lane(bend_1_road[0], bend_1_road[1], bend_1)
        self.net.add_lane(connect_road[0], connect_road[1], connect)
        ok = (not self.on_road(bend_1, 0.95)) and ok
        ok = (not self.on_road(connect, 0.95)) and ok
        bend_2, acc_lane = create_bend_straight(connect, acc_len, self.RADIUS, np.deg2rad(self.ANGLE_RAMP), True,
                                                self.lane_width, LT, speed_limit=12)
        acc_lane.line_types = [BROKEN, CONTINUOUS]
        bend_2_road = (connect_road[1], self.node(0, 0))
        self.net.add_lane(bend_2_road[0], bend_2_road[1], bend_2)
        self.net.add_lane(acc_road[0], acc_road[1], acc_lane)
        ok = (not self.on_road(bend_2, 0.95)) and ok
        ok = (not self.on_road(acc_lane, 0.95)) and ok
        merge, _ = create_bend_straight(acc_lane, 10, self.lane_width / 2, np.pi / 2, False, self.lane_width,
                                        (BROKEN, CONTINUOUS))
        self.net.add_lane(DEC_START, DEC_END, merge)
        return ok


class OutRamp(Block):
    ID = "R"
    EXTRA_LEN = 15

    def build(self):
        ok = True
        sa, ca = math.sin(np.deg2rad(self.ANGLE_RAMP)), math.cos(np.deg2rad(self.ANGLE_RAMP))
        longitude = sa * self.RADIUS * 2 + ca * self.CONNECT_PART_LEN + self.RAMP_LEN + self.EXTRA_LEN
        n = self.pos_lane_num
        self.set_part(0)
        dec_len = self.config["length"]
        dec_lane = extend_straight(self.pos_basic, dec_len + self.lane_width, [self.pos_basic.line_types[0], SIDE])
        dec_road = (self.pre_socket.pos[1], self.add_node())
        ok = self.rf(dec_lane, n, dec_road, side_type=CONTINUOUS) and ok
        ok = self.ar(dec_road) and ok
        dec_right = self.net.lanes(dec_road)[-1]
        dec_right.line_types = [CONTINUOUS if n == 1 else BROKEN, NONE]
        extend_lane = extend_straight(dec_right, longitude, [dec_right.line_types[0], CONTINUOUS])
        extend_road = (dec_road[1], self.add_node())
        ok = self.rf(extend_lane, n, extend_road, side_type=CONTINUOUS) and ok
        ok = self.ar(extend_road) and ok
        self.net.lanes(neg(extend_road))[-1].line_types = [NONE if n == 1 else BROKEN, SIDE]
        self.add_socket(self.socket_from_positive(extend_road))
        self.set_part(1)
        w = self.lane_width
        dec_side = SLane(dec_right.position(w, w), dec_right.position(dec_right.length, w), w, (BROKEN, CONTINUOUS))
        self.net.add_lane(dec_road[0], dec_road[1], dec_side)
        ok = (not self.on_road(dec_side, 0.95)) and ok
        LT = (CONTINUOUS, CONTINUOUS)
        bend_1, connect = create_bend_straight(dec_side, self.CONNECT_PART_LEN, self.RADIUS, np.deg2rad(self.ANGLE_RAMP), True,
                                               w, LT, speed_limit=12)
        bend_1_road = (dec_road[1], self.add_node())
        connect_road = (bend_1_road[1], self.add_node())
        self.net.add_lane(bend_1_road[0], bend_1_road[1], bend_1)
        self.net.add_lane(connect_road[0], connect_road[1], connect)
        ok = (not self.on_road(bend_1, 0.95)) and ok
        ok = (not self.on_road(connect, 0.95)) and ok
        bend_2, straight_part = create_bend_straight(connect, self.RAMP_LEN, self.RADIUS, np.deg2rad(self.ANGLE_RAMP), False, w,
                                                     LT, speed_limit=12)
        bend_2_road = (connect_road[1], self.add_node())
        straight_road = (bend_2_road[1], self.add_node())
        self.net.add_lane(bend_2_road[0], bend_2_road[1], bend_2)
        self.net.add_lane(straight_road[0], straight_road[1], straight_part)
        ok = (not self.on_road(bend_2, 0.95)) and ok
        ok = (not self.on_road(straight_part, 0.95)) and ok
        tool = SLane(dec_side.end, dec_side.start, dec_side.width)
        merge, _ = create_bend_straight(tool, 10, w / 2, np.pi / 2, True, width=w, line_types=(CONTINUOUS, BROKEN))
        self.net.add_lane(DEC_START, DEC_END, merge)
        return ok


class Intersection(Block):
    """StdInterSection (intersection.py:15-238 with change_lane_num forced to 0, std_intersection.py:5-9)"""
    ID = "X"
    EXIT_PART_LENGTH = 30
    enable_u_turn = False

    def build(self):
        self.config["change_lane_num"] = 0
        return self.build_x()

    def add_u_turn(self, enable):
        self.enable_u_turn = enable

    def build_x(self):
        p = self.config
        di = -1 if p["decrease_increase"] == 0 else 1
        if self.pos_lane_num <= 1:
            di = 1
        elif self.pos_lane_num >= 4:
            di = -1
        self.n_int = self.pos_lane_num + di * p["change_lane_num"]
        ok = True
        attach = self.pre_socket.pos
        attach_lanes = self.gnet.lanes(attach)
        nodes = deque([self.node(0, 0), self.node(1, 0), self.node(2, 0), self.pre_socket.neg[0]])
        for i in range(4):
            right_lane, good = self._part(attach_lanes, attach, p["radius"], nodes, i)
            ok = ok and good
            if i != 3:
                n = self.pos_lane_num if i == 1 else self.n_int
                exit_road = (self.node(i, 0), self.node(i, 1))
                ok = self.rf(right_lane, n, exit_road) and ok
                ok = self.ar(exit_road) and ok
                s = Socket(exit_road, neg(exit_road))
                self.respawn_roads.append(s.neg)
                self.add_socket(s)
                attach = neg(exit_road)
                attach_lanes = self.net.lanes(attach)
        return ok

    def _part(self, attach_lanes, attach, radius, nodes, part):
        n = self.n_int if part in (0, 2) else self.pos_lane_num
        good = True
        left = attach_lanes[0]
        self._left_turn(radius, n, left, attach, nodes, part)
        if self.enable_u_turn:  # InterSection._create_u_turn (intersection.py:207-230)
            bend, _ = create_bend_straight(left, 0.1, self.lane_width / 2, np.deg2rad(180), False, left.width, (NONE, NONE))
            self.rf(bend, len(attach_lanes), (attach[1], neg(attach)[0]), toward_smaller=False, center_line_type=NONE,
                    side_type=NONE, inner_type=NONE)
        on_road = copy.deepcopy(attach_lanes)
        straight_len = 2 * radius + (2 * n - 1) * on_road[0].width
        for l in on_road:
            self.net.add_lane(attach[1], nodes[1], extend_straight(l, straight_len, (NONE, NONE)))
        right_turn = on_road[-1]
        bend, right_straight = create_bend_straight(right_turn, self.EXIT_PART_LENGTH, radius, np.deg2rad(90), True,
                                                    right_turn.width, (NONE, SIDE))
        good = (not check_lane_on_road(self.gnet, bend, 1, skip=self.skip)) and good
        self.rf(bend, min(self.pos_lane_num, self.n_int), (attach[1], nodes[0]), toward_smaller=True, side_type=SIDE,
                inner_type=NONE, center_line_type=NONE)
        nodes.rotate(-1)
        right_straight.line_types = [BROKEN, SIDE]
        return right_straight, good

    def _left_turn(self, radius, n, left, attach, nodes, part):
        r = radius + n * left.width
        diff = self.n_int - self.pos_lane_num
        kw = dict(toward_smaller=False, center_line_type=NONE, side_type=NONE, inner_type=NONE)
        m = min(self.pos_lane_num, self.n_int)
        if ((part in (1, 3)) and diff > 0) or ((part in (0, 2)) and diff < 0):
            diff = abs(diff)
            bend, extra = create_bend_straight(left, self.lane_width * diff, r, np.deg2rad(90), False, left.width, (NONE, NONE))
            start = nodes[2]
            pre = start + "extra"
            self.rf(bend, m, (attach[1], pre), **kw)
            self.rf(extra, m, (pre, start), **kw)
        else:
            bend, _ = create_bend_straight(left, self.EXIT_PART_LENGTH, r, np.deg2rad(90), False, left.width, (NONE, NONE))
            self.rf(bend, m, (attach[1], nodes[2]), **kw)

    def get_socket(self, i):
        s = super().get_socket(i)
        if s.neg in self.respawn_roads:
            self.respawn_roads.remove(s.neg)
        return s

    def intermediate_spawn_lanes(self):
        return self.respawn_lanes()


class TIntersection(Intersection):
    """StdTInterSection (t_intersection.py:8-104, std_t_intersection.py:5-9)"""
    ID = "T"

    def build(self):
        self.config["change_lane_num"] = 0
        ok = self.build_x()
        self._exclude()
        return ok

    def _exclude(self):
        t = self.config["t_type"]
        self.add_socket(self.pre_socket)
        mine = self.sockets[real_index(self.name, t)]
        start_node, end_node = mine.neg[1], mine.pos[0]
        for i in range(4):
            if i == t:
                continue
            s = self.sockets[real_index(self.name, i) if i < 3 else self.pre_socket_index]
            exit_node = s.pos[0] if i != 3 else s.neg[0]
            self.net.remove_all_roads(start_node, exit_node)
            entry_node = s.neg[1] if i != 3 else s.pos[1]
            self.net.remove_all_roads(entry_node, end_node)
        self._change_vis(t)
        self.sockets.pop(self.pre_socket.index)
        s = self.sockets.pop(real_index(self.name, t))
        self.net.remove_all_roads(s.pos[0], s.pos[1])
        self.net.remove_all_roads(s.neg[0], s.neg[1])
        self.respawn_roads.remove(s.neg)

    def _change_vis(self, t):
        sl = list(self.sockets.values())
        nxt, last = sl[(t + 1) % 4], sl[(t + 3) % 4]
        next_pos, next_neg = nxt.pos, nxt.neg
        last_pos, last_neg = last.pos, last.neg
        if t == 2:  # Goal.LEFT
            next_pos, next_neg = nxt.neg, nxt.pos
        if t == 0:  # Goal.RIGHT
            last_pos, last_neg = last.neg, last.pos
        for i, road in enumerate([(last_neg[1], next_pos[0]), (next_neg[1], last_pos[0])]):
            lanes = self.net.lanes(road)
            outside = SIDE if i == 0 else NONE
            for k, lane in enumerate(lanes):
                lane.line_types = [BROKEN, BROKEN] if k != len(lanes) - 1 else [BROKEN, outside]
                if k == 0:
                    lane.line_color = [YELLOW, GREY]
                    if i == 1:
                        lane.line_types[0] = NONE


class Roundabout(Block):
    ID = "O"
    EXIT_PART_LENGTH = 30

    def build(self):
        self.mid_spawn = []
        p = self.config
        ok = True
        attach = self.pre_socket.pos
        for i in range(4):
            exit_road, good = self._part(attach, i, p["exit_radius"], p["inner_radius"], p["angle"])
            ok = ok and good
            if i < 3:
                ok = self.ar(exit_road) and ok
                attach = neg(exit_road)
        self.respawn_roads += [s.neg for s in self.sockets.values()]
        return ok

    def _part(self, road, part, r_exit, r_inner, angle):
        ok = True
        self.set_part(part)
        n = self.pos_lane_num
        w = self.lane_width
        r_big = (n * 2 - 1) * w + r_inner
        seg_start, seg_end = road[1], self.add_node()
        seg = (seg_start, seg_end)
        lanes = self.gnet.lanes(road) if part == 0 else self.net.lanes(road)
        bend, straight = create_bend_straight(lanes[-1], 10, r_exit, np.deg2rad(angle), True, w, (BROKEN, SIDE))
        ign = (self.node((part + 3) % 4, 0), self.node((part + 3) % 4, 0))
        ok = self.rf(bend, n, seg, ignore=ign) and ok
        for k, lane in enumerate(self.net.lanes(seg)):
            lane.line_types = [NONE, SIDE] if k == n - 1 else [NONE, NONE]
        tool = SLane(straight.position(-5, 0), straight.position(0, 0))
        bend, straight_next = create_bend_straight(tool, 10, r_big, np.deg2rad(2 * angle - 90), False, w, (BROKEN, SIDE))
        seg = (seg_end, self.add_node())
        ok = self.rf(bend, n, seg) and ok
        self.mid_spawn.append(self.net.lanes(seg))
        tool = SLane(straight_next.position(-5, 0), straight_next.position(0, 0))
        bend, straight = create_bend_straight(tool, self.EXIT_PART_LENGTH, r_exit, np.deg2rad(angle), True, w, (BROKEN, SIDE))
        seg = (seg[1], self.add_node() if part < 3 else self.pre_socket.neg[0])
        ok = self.rf(bend, n, seg) and ok
        for k, lane in enumerate(self.net.lanes(seg)):
            lane.line_types = [NONE, SIDE] if k == n - 1 else [NONE, NONE]
        exit_road = (seg[1], self.add_node())
        if part < 3:
            ok = self.rf(straight, n, exit_road) and ok
            self.add_socket(self.socket_from_positive(exit_road))
        seg = (self.node(part, 1), self.node((part + 1) % 4, 0))
        tool = SLane(straight_next.position(-6, 0), straight_next.position(0, 0))
        beneath = (n * 2 - 1) * w / 2 + r_exit
        r_this = beneath / math.cos(np.deg2rad(angle)) - r_exit
        bend, _ = create_bend_straight(tool, 5, r_this, np.deg2rad(180 - 2 * angle), False, w, (BROKEN, SIDE))
        self.rf(bend, n, seg)
        for k, lane in enumerate(self.net.lanes(seg)):
            if k == 0:
                lane.line_types = [CONTINUOUS, BROKEN] if n > 1 else [CONTINUOUS, NONE]
            else:
                lane.line_types = [BROKEN, BROKEN]
        return exit_road, ok

    def get_socket(self, i):
        s = super().get_socket(i)
        if s.neg in self.respawn_roads:
            self.respawn_roads.remove(s.neg)
        return s

    def intermediate_spawn_lanes(self):
        return self.respawn_lanes() + self.mid_spawn


# BLOCK_TYPE_DISTRIBUTION_V2 in dict order (blocks_prob_dist.py:31-49); zero-probability types keep their slot
class Merge(Block):
    """bottleneck.py:21-161: the outer `lane_num` lanes of each direction bend into the remaining straight ones."""
    ID = "y"

    def build(self):
        p = self.config
        L, changed = p["bottle_len"], p["lane_num"]
        start = self.pre_socket.pos[1]
        straight_n = max(1, int(self.pos_lane_num - changed))
        none3 = dict(center_line_type=NONE, inner_type=NONE)
        ref = extend_straight(self.pos_lanes[straight_n - 1], L, [NONE, NONE])
        sroad = (start, self.node(0, 0))
        ok = self.rf(ref, straight_n, sroad, center_line_type=CONTINUOUS, side_type=NONE, inner_type=NONE)
        ok = self.ar(sroad, inner_type=NONE, side_type=NONE, center_line_type=CONTINUOUS) and ok
        ref = extend_straight(ref, p["length"], [NONE, NONE])
        sock = (self.node(0, 0), self.node(0, 1))
        ok = self.rf(ref, straight_n, sock, center_line_type=CONTINUOUS, side_type=SIDE, inner_type=BROKEN) and ok
        ok = self.ar(sock, inner_type=BROKEN, side_type=SIDE, center_line_type=CONTINUOUS) and ok
        self.add_socket(Socket(sock, neg(sock)))
        for index, lane in enumerate(self.pos_lanes[straight_n:], 1):
            lat = index * self.lane_width / 2
            inner = self.node(1, index)
            side_t = SIDE if index == self.pos_lane_num - straight_n else NONE
            c1, c2, _ = create_wave_lanes(lane, lat, L, 5, self.lane_width)
            r1, r2 = (start, inner), (inner, self.node(0, 0))
            ok = self.rf(c1, 1, r1, side_type=side_t, **none3) and ok
            ok = self.rf(c2, 1, r2, side_type=side_t, **none3) and ok
            lane2 = self.net.lanes(neg(sock))[-1]
            c2b, c1b, _ = create_wave_lanes(lane2, lat, L, 5, self.lane_width, False)
            ok = self.rf(c2b, 1, neg(r2), side_type=side_t, **none3) and ok
            ok = self.rf(c1b, 1, neg(r1), side_type=side_t, **none3) and ok
        return ok


class Split(Block):
    """bottleneck.py:164-311: `lane_num` extra lanes fan out on each side after the neck."""
    ID = "Y"

    def build(self):
        p = self.config
        L, circ_n = p["bottle_len"], p["lane_num"]
        start = self.pre_socket.pos[1]
        straight_n = self.pos_lane_num
        total = straight_n + circ_n
        none3 = dict(center_line_type=NONE, inner_type=NONE)
        ref = extend_straight(self.pos_lanes[straight_n - 1], L, [NONE, NONE])
        sroad = (start, self.node(0, 0))
        ok = self.rf(ref, straight_n, sroad, center_line_type=CONTINUOUS, side_type=NONE, inner_type=NONE)
        ok = self.ar(sroad, inner_type=NONE, side_type=NONE, center_line_type=CONTINUOUS) and ok
        lane = self.pos_lanes[-1]
        sock_ref = None
        for index in range(1, circ_n + 1):
            lat = index * self.lane_width / 2
            inner = self.node(1, index)
            side_t = SIDE if index == circ_n else NONE
            c1, c2, straight = create_wave_lanes(lane, lat, L, p["length"], self.lane_width, False)
            if index == circ_n:
                sock_ref = straight
            ok = self.rf(c1, 1, (start, inner), side_type=side_t, **none3) and ok
            ok = self.rf(c2, 1, (inner, self.node(0, 0)), side_type=side_t, **none3) and ok
        sock = (self.node(0, 0), self.node(0, 1))
        ok = self.rf(sock_ref, total, sock, center_line_type=CONTINUOUS, side_type=SIDE, inner_type=BROKEN) and ok
        ok = self.ar(sock, inner_type=BROKEN, side_type=SIDE, center_line_type=CONTINUOUS) and ok
        self.add_socket(Socket(sock, neg(sock)))
        lanes = self.net.lanes(neg(sock))
        for index, lane in enumerate(lanes[self.pos_lane_num:], 1):
            lat = index * self.lane_width / 2
            inner = self.node(1, index)
            side_t = SIDE if index == circ_n else NONE
            c1, c2, _ = create_wave_lanes(lane, lat, L, 5, self.lane_width)
            ok = self.rf(c1, 1, neg((inner, self.node(0, 0))), side_type=side_t, **none3) and ok
            ok = self.rf(c2, 1, neg((start, inner)), side_type=side_t, **none3) and ok
        return ok


class TollGate(Block):
    """tollgate.py:16-81: a straight piece whose lanes are separated by continuous lines, limited to 3 (sic: compared with
    km/h, base_vehicle.py:760-761), with a booth (invisible wall, 10 m x lane width) in the middle of every odd lane."""
    ID = "$"
    SPEED_LIMIT = 3
    BUILDING_LENGTH = 10  # TollGateBuilding.BUILDING_LENGTH (tollgate_building.py:8)

    def build(self):
        self.set_part(0)
        new_lane = extend_straight(self.pos_basic, self.config["length"], [CONTINUOUS, SIDE])
        sock = (self.pre_socket.pos[1], self.add_node())
        kw = dict(center_color=YELLOW, center_line_type=CONTINUOUS, inner_type=CONTINUOUS, side_type=SIDE)
        ok = self.rf(new_lane, self.pos_lane_num, sock, **kw)
        ok = self.ar(sock, **kw) and ok
        self.add_socket(Socket(sock, neg(sock)))
        self.buildings = []
        for road in (sock, neg(sock)):
            for idx, lane in enumerate(self.net.lanes(road)):
                lane.speed_limit = self.SPEED_LIMIT
                if idx % 2 == 1:
                    p = lane.position(lane.length / 2, 0)
                    self.buildings.append(dict(lane=lane, x=float(p[0]), y=float(p[1]), heading=float(lane.heading_at(0)),
                                               length=float(self.BUILDING_LENGTH), width=float(lane.width)))
        return ok


class ParkingLot(Block):
    """parking_lot.py:13-329: a one-lane two-way road with `one_side_vehicle_number` perpendicular parking spaces on each
    side; every space has an entry arc from either direction of the main road and exit arcs back to both directions, the
    space itself is a two-way road (in: nodes 1->2, out: nodes 5->6)."""
    ID = "P"
    ANGLE = np.deg2rad(90)
    SOCKET_LENGTH = 4

    def build(self):
        self.spawn_roads, self.dest_roads = [], []
        p = self.config
        assert self.pos_lane_num == 1, "Lane number of previous block must be 1 in each direction"
        self.space_length = p["length"]
        self.space_width = self.lane_width
        n = int(p["one_side_vehicle_number"])
        radius = p["radius"]
        main_len = 2 * radius + (n - 1) * self.space_width
        main_lane = extend_straight(self.pos_lanes[0], main_len, [BROKEN, NONE])
        road = (self.pre_socket.pos[1], self.node(0, 0))
        kw = dict(center_line_type=BROKEN, inner_type=BROKEN, side_type=NONE, center_color=GREY)
        ok = self.rf(main_lane, self.pos_lane_num, road, **kw)
        ok = self.ar(road, **kw) and ok
        out_lane = extend_straight(main_lane, self.SOCKET_LENGTH, [BROKEN, NONE])
        out_road = (self.node(0, 0), self.node(0, 1))
        kw2 = dict(center_line_type=BROKEN, inner_type=BROKEN, side_type=SIDE)
        ok = self.rf(out_lane, self.pos_lane_num, out_road, **kw2) and ok
        ok = self.ar(out_road, **kw2) and ok
        sock = Socket(out_road, neg(out_road))
        self.add_socket(sock)
        rev = lambda s: Socket(s.neg, s.pos)
        for i in range(n):
            ok = self._space(rev(sock), rev(self.pre_socket), i + 1, radius, i * self.space_width,
                             (n - i - 1) * self.space_width) and ok
        for i in range(n, 2 * n):
            j = i - n
            ok = self._space(self.pre_socket, sock, i + 1, radius, j * self.space_width, (n - j - 1) * self.space_width) and ok
        return ok

    def _is_pre(self, s):
        a, b = self.pre_socket.pos, self.pre_socket.neg
        return (s.pos == a and s.neg == b) or (s.pos == b and s.neg == a)

    def _space(self, in_s, out_s, part, radius, dist_in, dist_out):
        ok = True
        none = dict(center_line_type=NONE, inner_type=NONE, side_type=NONE)
        net = self.gnet if self._is_pre(in_s) else self.net
        in_lane = net.lanes(in_s.pos)[0]
        start = in_s.pos[1]
        if dist_in > 1e-3:
            in_lane = extend_straight(in_lane, dist_in, [NONE, NONE])
            self.rf(in_lane, self.pos_lane_num, (in_s.pos[1], self.node(part, 0)), **none)
            start = self.node(part, 0)
        bend, straight = create_bend_straight(in_lane, self.space_length, radius, self.ANGLE, True, self.space_width)
        side_in = SIDE if dist_in < 1e-3 else NONE
        bend_ok = self.rf(bend, self.pos_lane_num, (start, self.node(part, 1)), center_line_type=NONE, inner_type=NONE,
                          side_type=side_in)
        if dist_in < 1e-3:
            ok = ok and bend_ok
        space_road = (self.node(part, 1), self.node(part, 2))
        self.dest_roads.append(space_road)
        ok = ok and self.rf(straight, self.pos_lane_num, space_road, center_line_type=CONTINUOUS, inner_type=NONE,
                            side_type=side_in, center_color=GREY)
        # the second way in, from the other direction of the main road
        nroad = out_s.neg
        net = self.gnet if self._is_pre(out_s) else self.net
        nlane = net.lanes(nroad)[0]
        start = nroad[1]
        if dist_out > 1e-3:
            nlane = extend_straight(nlane, dist_out, [NONE, NONE])
            self.rf(nlane, self.pos_lane_num, (nroad[1], self.node(part, 3)), **none)
            start = self.node(part, 3)
        bend, straight = create_bend_straight(nlane, self.lane_width, radius, self.ANGLE, False, self.space_width)
        self.rf(bend, self.pos_lane_num, (start, self.node(part, 4)), **none)
        self.rf(straight, self.pos_lane_num, (self.node(part, 4), self.node(part, 1)), **none)
        # the space driven outwards is the two-way twin (5 -> 6) of (1 -> 2)
        park_road = (self.node(part, 5), self.node(part, 6))
        self.spawn_roads.append(park_road)
        side_out = SIDE if dist_out < 1e-3 else NONE
        create_two_way_road(space_road, self.net, self.gnet, park_road, center_line_type=NONE, inner_type=NONE,
                            side_type=side_out, skip=self.skip)
        park_lane = self.net.lanes(park_road)[0]
        bend, straight = create_bend_straight(park_lane, 0.1 if dist_out < 1e-3 else dist_out, radius, self.ANGLE, True,
                                              park_lane.width)
        out_bend = (self.node(part, 6), self.node(part, 7) if dist_out > 1e-3 else out_s.pos[0])
        bend_ok = self.rf(bend, self.pos_lane_num, out_bend, center_line_type=NONE, inner_type=NONE, side_type=side_out)
        if dist_out < 1e-3:
            ok = ok and bend_ok
        if dist_out > 1e-3:
            ok = ok and self.rf(straight, self.pos_lane_num, (self.node(part, 7), out_s.pos[0]), **none)
        ext = extend_straight(park_lane, self.lane_width, [NONE, NONE])
        self.rf(ext, self.pos_lane_num, (self.node(part, 6), self.node(part, 8)), **none)
        bend, straight = create_bend_straight(ext, 0.1 if dist_in < 1e-3 else dist_in, radius, self.ANGLE, False,
                                              park_lane.width)
        out_bend = (self.node(part, 8), self.node(part, 9) if dist_in > 1e-3 else in_s.neg[0])
        self.rf(bend, self.pos_lane_num, out_bend, **none)
        if dist_in > 1e-3:
            self.rf(straight, self.pos_lane_num, (self.node(part, 9), in_s.neg[0]), **none)
        return ok


BLOCK_TYPES = [Curve, Straight, InRamp, OutRamp, Intersection, TIntersection, Roundabout, None, None, None, None, None, None]
BLOCK_PROBS = [0.3, 0.1, 0.1, 0.1, 0.15, 0.15, 0.1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
BY_ID = {c.ID: c for c in BLOCK_TYPES if c is not None}


# ----------------------------------------------------------------------------------------------------------------------
# BIG (component/algorithm/BIG.py)
# ----------------------------------------------------------------------------------------------------------------------
def generate_blocks(seed, lane_num=3, lane_width=3.5, exit_length=50, block_num=None, block_seq=None):
    rng = get_np_random(seed)
    gnet = Net()
    blocks = [FirstBlock(gnet, lane_width, lane_num, exit_length)]
    if block_seq is not None:
        target = len(block_seq) + 1
        seq = "I" + block_seq
    else:
        target, seq = block_num + 1, None
    FORWARD, DESTRUCT, SIBLING, BACK = 1, 4, 3, 0
    step = FORWARD

    def sample_block():
        if seq is None:
            cls = BLOCK_TYPES[int(rng.choice(len(BLOCK_TYPES), p=BLOCK_PROBS))]
        else:
            cls = BY_ID[seq[len(blocks)]]
        sock = rng.choice(blocks[-1].socket_indices())
        return cls(len(blocks), blocks[-1].get_socket(sock), gnet, int(rng.randint(0, 10000)))

    while True:
        if len(blocks) >= target and step == FORWARD:
            break
        if step == FORWARD:
            b = sample_block()
            blocks.append(b)
            step = FORWARD if b.construct() else DESTRUCT
        elif step == DESTRUCT:
            b = blocks[-1]
            b.destruct()
            step = SIBLING if b.trials < 2 else BACK
        elif step == SIBLING:
            b = blocks[-1]
            if b.trials < 2:
                step = FORWARD if b.construct() else DESTRUCT
            else:
                step = BACK
        else:
            blocks.pop()
            blocks[-1].destruct()
            step = SIBLING
    return gnet, blocks


# ----------------------------------------------------------------------------------------------------------------------
# map files: the reference's own JSON (BaseMap.save_map / read_map, component/map/base_map.py:103-130;
# PGMap._config_generate, pg_map.py:51-80; PGDriveEnv.dump_all_maps / load_all_maps, envs/pgdrive_env.py:260-330)
# ----------------------------------------------------------------------------------------------------------------------
def save_map(blocks):
    """BaseMap.save_map: {"block_sequence": [block parameters + "id" + "pre_block_socket_index"]}."""
    seq = []
    for b in blocks:
        cfg = dict(b.config)
        cfg["id"] = b.ID
        cfg["pre_block_socket_index"] = b.pre_socket_index
        seq.append(cfg)
    return {"block_sequence": seq}


def generate_from_block_sequence(block_sequence, seed=0, lane_num=3, lane_width=3.5, exit_length=50):
    """PGMap._config_generate: rebuild a map from its saved block sequence (every block takes its parameters from the
    file, crossing checks are skipped like ignore_intersection_checking=True)."""
    by_id = dict(BY_ID)
    by_id.update({"y": Merge, "Y": Split, "$": TollGate, "P": ParkingLot})
    gnet = Net()
    blocks = [FirstBlock(gnet, lane_width, lane_num, exit_length)]
    for k, b in enumerate(block_sequence[1:], 1):
        cfg = {key: (v if not isinstance(v, list) else np.array(v)) for key, v in b.items() if key not in ("id", "pre_block_socket_index")}
        blk = by_id[b["id"]](k, blocks[-1].get_socket(b["pre_block_socket_index"]), gnet, seed, skip_check=True)
        blk.construct(extra_config=cfg)
        blocks.append(blk)
    return to_description(seed, gnet, blocks, lane_num, lane_width, exit_length)


def dump_all_maps(start_seed, environment_num, lane_num=3, lane_width=3.5, exit_length=50, block_num=3, block_seq=None):
    """PGDriveEnv.dump_all_maps: {"map_config": {...}, "map_data": {seed: save_map()}} (JSON-serialisable)."""
    data = {}
    for seed in range(start_seed, start_seed + environment_num):
        _, blocks = generate_blocks(seed, lane_num, lane_width, exit_length, block_num, block_seq)
        data[seed] = save_map(blocks)
    mc = dict(lane_num=lane_num, lane_width=lane_width, exit_length=exit_length,
              type="block_num" if block_seq is None else "block_sequence", config=block_num if block_seq is None else block_seq)
    return dict(map_config=mc, map_data=data)


def load_all_maps(data):
    """PGDriveEnv.load_all_maps: descriptions for every seed of a dump (ours or the reference's own file)."""
    mc = data["map_config"]
    out = []
    for seed, m in sorted(data["map_data"].items(), key=lambda kv: int(kv[0])):
        out.append(generate_from_block_sequence(m["block_sequence"], int(seed), mc.get("lane_num", 3), mc.get("lane_width", 3.5),
                                                mc.get("exit_length", 50)))
    return out


def to_description(seed, gnet, blocks, lane_num, lane_width, exit_length):
    """Flatten to the description format of the map bank (same keys as the bank exported from the reference)."""
    nodes = []

    def nid(n):
        if n not in nodes:
            nodes.append(n)
        return nodes.index(n)

    lane_id, lanes, roads = {}, [], []
    for a, td in gnet.graph.items():
        for b, ls in td.items():
            valid = is_valid((a, b))
            if valid:
                search = b if not is_negative((a, b)) else a
                bid = ">" if ">" in search else next(ch for ch in search if ch.isalpha() or ch == "$")
            else:
                bid = "?"
            roads.append(dict(frm=nid(a), to=nid(b), first_lane=len(lanes), n_lanes=len(ls), negative=is_negative((a, b)),
                              block_id=bid, valid=valid))
            for i, l in enumerate(ls):
                lane_id[id(l)] = len(lanes)
                d = dict(road=len(roads) - 1, index=i, length=float(l.length), width=float(l.width),
                         line_types=[int(l.line_types[0]), int(l.line_types[1])],
                         line_colors=[int(l.line_color[0]), int(l.line_color[1])],
                         start=[float(l.start[0]), float(l.start[1])], end=[float(l.end[0]), float(l.end[1])])
                if l.kind == 0:
                    d.update(type=0, heading=float(l.heading), direction=[float(l.direction[0]), float(l.direction[1])])
                else:
                    d.update(type=1, center=[float(l.center[0]), float(l.center[1])], radius=float(l.radius),
                             start_phase=float(l.start_phase), end_phase=float(l.end_phase), direction=int(l.direction))
                lanes.append(d)
    rl = {(r["frm"], r["to"]): i for i, r in enumerate(roads)}
    out_blocks = []
    for b in blocks:
        sockets = [dict(pos=[nid(s.pos[0]), nid(s.pos[1])], neg=[nid(s.neg[0]), nid(s.neg[1])]) for s in b.sockets.values()]
        spawn = [[lane_id[id(l)] for l in ls] for ls in b.intermediate_spawn_lanes()]
        broads = [[rl[(nid(a), nid(t))], [lane_id[id(l)] for l in ls]] for a, td in b.net.graph.items() for t, ls in td.items()]
        trig = b.pre_socket.pos if b.index != 0 else None
        extra = {}
        if getattr(b, "buildings", None):
            extra["buildings"] = [dict(lane=lane_id[id(x["lane"])], x=x["x"], y=x["y"], heading=x["heading"], length=x["length"],
                                       width=x["width"]) for x in b.buildings]
        out_blocks.append(dict(id=b.ID, sockets=sockets, spawn_lanes=spawn, **extra,
                               respawn_roads=[[nid(r[0]), nid(r[1])] for r in b.respawn_roads], roads=broads,
                               trigger_road=[nid(trig[0]), nid(trig[1])] if trig else None,
                               config=dict(b.config) if b.index != 0 else {}, pre_socket=b.pre_socket_index))
    return dict(seed=seed, lane_num=lane_num, lane_width=lane_width, exit_length=exit_length, nodes=nodes, roads=roads,
                lanes=lanes, blocks=out_blocks)


def generate_map(seed, lane_num=3, lane_width=3.5, exit_length=50, block_num=3, block_seq=None):
    """PGMap._big_generate (component/map/pg_map.py:34-46) -> map description."""
    gnet, blocks = generate_blocks(seed, lane_num, lane_width, exit_length, None if block_seq else block_num, block_seq)
    return to_description(seed, gnet, blocks, lane_num, lane_width, exit_length)


class FullIntersection(Intersection):
    """InterSection proper (intersection.py:15-238): the lane-count change of the crossing road is sampled, not forced to 0."""
    def build(self):
        return self.build_x()


def generate_ma_parking_lot(lane_width=3.5, exit_length=20, parking_space_num=8):
    """MAParkingLotMap._generate (envs/marl_envs/marl_parking_lot.py:92-130): one-lane first block, ParkingLot with
    parking_space_num / 2 spaces per side, T-intersection (t_type 1, exits 10 m)."""
    gnet = Net()
    first = FirstBlock(gnet, lane_width, 1, exit_length)
    lot = ParkingLot(1, first.get_socket(0), gnet, 1)
    assert lot.construct(extra_config={"one_side_vehicle_number": int(parking_space_num / 2)})
    t = TIntersection(2, lot.get_socket(0), gnet, 1)
    t.EXIT_PART_LENGTH = 10
    assert t.construct(extra_config={"t_type": 1, "change_lane_num": 0})
    return to_description(0, gnet, [first, lot, t], 1, lane_width, exit_length)


def generate_ma_tollgate(lane_num=3, lane_width=3.5, exit_length=70, toll_lane_num=8, toll_length=10, bottle_length=35):
    """MATollGateMap._generate (envs/marl_envs/marl_tollgate.py:108-160): 3-lane first block, Split out to 8 lanes, the toll
    plaza, Merge back to 3 lanes."""
    gnet = Net()
    first = FirstBlock(gnet, lane_width, lane_num, exit_length)
    split = Split(1, first.get_socket(0), gnet, 1)
    assert split.construct(extra_config=dict(length=2, lane_num=toll_lane_num - lane_num, bottle_len=bottle_length))
    toll = TollGate(2, split.get_socket(0), gnet, 1)
    assert toll.construct(extra_config=dict(length=toll_length))
    merge = Merge(3, toll.get_socket(0), gnet, 1)
    assert merge.construct(extra_config=dict(lane_num=toll_lane_num - lane_num, length=exit_length, bottle_len=bottle_length))
    return to_description(0, gnet, [first, split, toll, merge], lane_num, lane_width, exit_length)


def generate_ma_bottleneck(lane_width=3.5, exit_length=60, bottle_lane_num=4, neck_lane_num=1, neck_length=20):
    """MABottleneckMap._generate (envs/marl_envs/marl_bottleneck.py:28-67): 4-lane first block, Merge down to the neck,
    Split back to 4 lanes."""
    gnet = Net()
    first = FirstBlock(gnet, lane_width, bottle_lane_num, exit_length)
    merge = Merge(1, first.get_socket(0), gnet, 1)
    assert merge.construct(extra_config=dict(lane_num=bottle_lane_num - neck_lane_num, length=neck_length))
    split = Split(2, merge.get_socket(0), gnet, 1)
    assert split.construct(extra_config=dict(length=exit_length, lane_num=bottle_lane_num - neck_lane_num))
    return to_description(0, gnet, [first, merge, split], bottle_lane_num, lane_width, exit_length)


def generate_ma_intersection(lane_num=2, lane_width=3.5, exit_length=60):
    """MAIntersectionMap._generate (envs/marl_envs/marl_intersection.py:29-54): first block + one intersection (block
    seed 1) with u-turns and exit parts as long as the entrance road."""
    gnet = Net()
    first = FirstBlock(gnet, lane_width, lane_num, exit_length)
    x = FullIntersection(1, first.get_socket(0), gnet, 1)
    x.EXIT_PART_LENGTH = exit_length
    x.add_u_turn(True)
    ok = x.construct()
    assert ok
    return to_description(0, gnet, [first, x], lane_num, lane_width, exit_length)


def generate_ma_roundabout(lane_num=2, lane_width=3.5, exit_length=60):
    """MARoundaboutMap._generate (envs/marl_envs/marl_inout_roundabout.py:30-63): first block + one roundabout with
    exit_radius 10, inner_radius 30, angle 70 and exit parts as long as the entrance road."""
    gnet = Net()
    first = FirstBlock(gnet, lane_width, lane_num, exit_length)
    rb = Roundabout(1, first.get_socket(0), gnet, 1)
    rb.EXIT_PART_LENGTH = exit_length
    ok = rb.construct(extra_config={"exit_radius": 10, "inner_radius": 30, "angle": 70})
    assert ok
    return to_description(0, gnet, [first, rb], lane_num, lane_width, exit_length)
