"""Export map descriptions + box tables from the *reference's own* BIG generator (TEST INFRASTRUCTURE).

Runs only in the build container (needs /root/reference).  Produces neutral, data-only descriptions:
lanes (geometry, line types/colours, road, index), roads in `RoadNetwork.graph` insertion order, block metadata
(sockets, trigger road, spawn lanes, respawn roads) and the OBB tables the reference would have handed to Bullet
(`pgdrive/component/blocks/base_block.py:142-464`), recorded through the stubs in `refstub.py`.
"""
import math
import types

import numpy as np

import refstub

refstub.load()

from pgdrive.component.algorithm.BIG import BIG, BigGenerateMethod  # noqa: E402
from pgdrive.component.blocks.base_block import BaseBlock  # noqa: E402
from pgdrive.component.lane.circular_lane import CircularLane  # noqa: E402
from pgdrive.component.lane.straight_lane import StraightLane  # noqa: E402
from pgdrive.component.road.road import Road  # noqa: E402
from pgdrive.component.road.road_network import RoadNetwork  # noqa: E402
from pgdrive.constants import BodyName, LineColor, LineType  # noqa: E402

KIND = {
    BodyName.Lane: 0,
    BodyName.White_continuous_line: 1,
    BodyName.Yellow_continuous_line: 2,
    BodyName.Broken_line: 3,
    BodyName.Sidewalk: 4,
}
LINE_TYPE_CODE = {LineType.NONE: 0, LineType.BROKEN: 1, LineType.CONTINUOUS: 2, LineType.SIDE: 3}

# ---- per-block recording of created boxes: wrap _create_in_world so destructed blocks drop their boxes ----
_orig_create = BaseBlock._create_in_world


def _rec_create(self):
    start = len(refstub.RECORD)
    _orig_create(self)
    self._rec_boxes = refstub.RECORD[start:]
    del refstub.RECORD[start:]


BaseBlock._create_in_world = _rec_create


def _decode_box(np_):
    node = np_.node()
    name = node.getName()
    kind = KIND[name]
    half = node.shapes[0].half
    q = np_.quat
    t = 2.0 * math.atan2(q[3], q[0])  # panda heading; pgdrive heading = -t
    theta = -t
    x, y = np_.pos[0], -np_.pos[1]
    if kind == 0:
        hl, hw = half[0], half[2]  # lane surface box is rotated -90deg about x (base_block.py:446-456)
        lane = node.base_object_name
    elif kind == 4:
        hl, hw = half[0] * np_.scale[0], half[1] * np_.scale[1]
        lane = None
    else:
        hl, hw = half[0], half[1]
        lane = None
    return kind, x, y, theta, hl, hw, lane


def _color_code(c):
    return 1 if tuple(c) == tuple(LineColor.YELLOW) else 0


class _Blocks:
    def __init__(self, blocks):
        self.blocks = blocks


def generate_ma_roundabout(lane_num=2, lane_width=3.5, exit_length=60):
    """MARoundaboutMap._generate (envs/marl_envs/marl_inout_roundabout.py:30-63): first block + one roundabout with
    exit_radius 10, inner_radius 30, angle 70."""
    from pgdrive.component.blocks.first_block import FirstPGBlock
    from pgdrive.component.blocks.roundabout import Roundabout
    net = RoadNetwork()
    pw = refstub.FakePhysicsWorld()
    first = FirstPGBlock(net, lane_width, lane_num, None, pw, length=exit_length)
    Roundabout.EXIT_PART_LENGTH = exit_length
    rb = Roundabout(1, first.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
    ok = rb.construct_block(None, pw, extra_config={"exit_radius": 10, "inner_radius": 30, "angle": 70})
    assert ok
    return generate(0, lane_num, lane_width, exit_length, prebuilt=(net, _Blocks([first, rb])))


def generate_ma_parking_lot(lane_width=3.5, exit_length=20, parking_space_num=8):
    """MAParkingLotMap._generate (envs/marl_envs/marl_parking_lot.py:92-130)."""
    from pgdrive.component.blocks.first_block import FirstPGBlock
    from pgdrive.component.blocks.parking_lot import ParkingLot
    from pgdrive.component.blocks.t_intersection import TInterSection
    net = RoadNetwork()
    pw = refstub.FakePhysicsWorld()
    first = FirstPGBlock(net, lane_width, 1, None, pw, length=exit_length)
    lot = ParkingLot(1, first.get_socket(0), net, 1, ignore_intersection_checking=False)
    assert lot.construct_block(None, pw, {"one_side_vehicle_number": int(parking_space_num / 2)})
    old = TInterSection.EXIT_PART_LENGTH
    TInterSection.EXIT_PART_LENGTH = 10
    try:
        t = TInterSection(2, lot.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
        assert t.construct_block(None, pw, extra_config={"t_type": 1, "change_lane_num": 0})
    finally:
        TInterSection.EXIT_PART_LENGTH = old
    m = generate(0, 1, lane_width, exit_length, prebuilt=(net, _Blocks([first, lot, t])))
    m["parking_space"] = [[r.start_node, r.end_node] for r in lot.dest_roads]
    m["parking_spawn"] = [[r.start_node, r.end_node] for r in lot.spawn_roads]
    return m


def generate_ma_tollgate(lane_num=3, lane_width=3.5, exit_length=70, toll_lane_num=8, toll_length=10, bottle_length=35):
    """MATollGateMap._generate (envs/marl_envs/marl_tollgate.py:108-160); the toll booths spawned through get_engine() are
    recorded (lane, position, heading)."""
    from pgdrive.component.blocks import tollgate as tg
    from pgdrive.component.blocks.bottleneck import Merge, Split
    from pgdrive.component.blocks.first_block import FirstPGBlock
    net = RoadNetwork()
    pw = refstub.FakePhysicsWorld()
    booths = []

    def spawn_object(cls, lane=None, position=None, heading=None):
        booths.append(dict(lane=lane, x=float(position[0]), y=float(position[1]), heading=float(heading),
                           length=float(cls.BUILDING_LENGTH), width=float(lane.width)))
        return types.SimpleNamespace(body=None)
    old = tg.get_engine
    tg.get_engine = lambda: types.SimpleNamespace(spawn_object=spawn_object)
    try:
        first = FirstPGBlock(net, lane_width, lane_num, None, pw, length=exit_length)
        split = Split(1, first.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
        assert split.construct_block(None, pw, {"length": 2, "lane_num": toll_lane_num - lane_num, "bottle_len": bottle_length})
        toll = tg.TollGate(2, split.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
        assert toll.construct_block(None, pw, {"length": toll_length})
        merge = Merge(3, toll.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
        assert merge.construct_from_config(dict(lane_num=toll_lane_num - lane_num, length=exit_length, bottle_len=bottle_length),
                                           None, pw)
    finally:
        tg.get_engine = old
    m = generate(0, lane_num, lane_width, exit_length, prebuilt=(net, _Blocks([first, split, toll, merge])))
    lanes = []
    for _f, td in net.graph.items():
        for _t, ls in td.items():
            lanes += ls
    m["booths"] = [dict(lane=[k for k, l in enumerate(lanes) if l is b["lane"]][0], x=b["x"], y=b["y"], heading=b["heading"],
                        length=b["length"], width=b["width"]) for b in booths]
    return m


def generate_ma_bottleneck(lane_width=3.5, exit_length=60, bottle_lane_num=4, neck_lane_num=1, neck_length=20):
    """MABottleneckMap._generate (envs/marl_envs/marl_bottleneck.py:28-67)."""
    from pgdrive.component.blocks.bottleneck import Merge, Split
    from pgdrive.component.blocks.first_block import FirstPGBlock
    net = RoadNetwork()
    pw = refstub.FakePhysicsWorld()
    first = FirstPGBlock(net, lane_width, bottle_lane_num, None, pw, length=exit_length)
    merge = Merge(1, first.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
    assert merge.construct_from_config(dict(lane_num=bottle_lane_num - neck_lane_num, length=neck_length), None, pw)
    split = Split(2, merge.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
    assert split.construct_from_config({"length": exit_length, "lane_num": bottle_lane_num - neck_lane_num}, None, pw)
    return generate(0, bottle_lane_num, lane_width, exit_length, prebuilt=(net, _Blocks([first, merge, split])))


def generate_ma_intersection(lane_num=2, lane_width=3.5, exit_length=60):
    """MAIntersectionMap._generate (envs/marl_envs/marl_intersection.py:29-54): first block + one intersection with
    u-turns, exit parts as long as the entrance road."""
    from pgdrive.component.blocks.first_block import FirstPGBlock
    from pgdrive.component.blocks.intersection import InterSection
    net = RoadNetwork()
    pw = refstub.FakePhysicsWorld()
    first = FirstPGBlock(net, lane_width, lane_num, None, pw, length=exit_length)
    old = InterSection.EXIT_PART_LENGTH
    InterSection.EXIT_PART_LENGTH = exit_length
    try:
        x = InterSection(1, first.get_socket(index=0), net, random_seed=1, ignore_intersection_checking=False)
        x.add_u_turn(True)
        ok = x.construct_block(None, pw)
        assert ok
        return generate(0, lane_num, lane_width, exit_length, prebuilt=(net, _Blocks([first, x])))
    finally:
        InterSection.EXIT_PART_LENGTH = old


def generate(seed, lane_num=3, lane_width=3.5, exit_length=50, block_num=None, block_seq=None, prebuilt=None):
    """Run the reference BIG and flatten the result into plain python/numpy data."""
    if prebuilt is not None:
        net, big = prebuilt
    else:
        net = RoadNetwork()
        big = BIG(lane_num, lane_width, net, None, refstub.FakePhysicsWorld(), exit_length=exit_length, random_seed=seed)
        if block_seq is not None:
            big.generate(BigGenerateMethod.BLOCK_SEQUENCE, block_seq)
        else:
            big.generate(BigGenerateMethod.BLOCK_NUM, block_num)
    net.after_init()

    nodes = []  # node names in first-seen order

    def nid(n):
        if n not in nodes:
            nodes.append(n)
        return nodes.index(n)

    lane_id = {}
    lanes = []
    roads = []
    for _from, td in net.graph.items():
        for _to, ls in td.items():
            r = Road(_from, _to)
            roads.append(
                dict(
                    frm=nid(_from), to=nid(_to), first_lane=len(lanes), n_lanes=len(ls),
                    negative=bool(r.is_negative_road()), block_id=(r.block_ID() if r.is_valid_road() else "?"),
                    valid=bool(r.is_valid_road())
                )
            )
            for i, l in enumerate(ls):
                lane_id[id(l)] = len(lanes)
                d = dict(
                    road=len(roads) - 1, index=i, length=float(l.length), width=float(l.width),
                    line_types=[LINE_TYPE_CODE[l.line_types[0]], LINE_TYPE_CODE[l.line_types[1]]],
                    line_colors=[_color_code(l.line_color[0]), _color_code(l.line_color[1])],
                    start=[float(l.start[0]), float(l.start[1])], end=[float(l.end[0]), float(l.end[1])],
                )
                if isinstance(l, StraightLane):
                    d.update(type=0, heading=float(l.heading), direction=[float(l.direction[0]), float(l.direction[1])])
                elif isinstance(l, CircularLane):
                    d.update(
                        type=1, center=[float(l.center[0]), float(l.center[1])], radius=float(l.radius),
                        start_phase=float(l.start_phase), end_phase=float(l.end_phase), direction=int(l.direction)
                    )
                else:
                    raise TypeError(type(l))
                lanes.append(d)

    road_lookup = {(r["frm"], r["to"]): i for i, r in enumerate(roads)}
    boxes = []
    blocks = []
    for b in big.blocks:
        for np_ in b._rec_boxes:
            kind, x, y, theta, hl, hw, lane = _decode_box(np_)
            boxes.append((kind, x, y, theta, hl, hw, lane_id[id(lane)] if lane is not None else -1))
        sockets = [
            dict(
                pos=[nid(s.positive_road.start_node), nid(s.positive_road.end_node)],
                neg=[nid(s.negative_road.start_node), nid(s.negative_road.end_node)]
            ) for s in b.get_socket_list()
        ]
        spawn_lanes = [[lane_id[id(l)] for l in ls] for ls in b.get_intermediate_spawn_lanes()]
        rl = road_lookup
        broads = []
        for _f, td in b.block_network.graph.items():
            for _t, ls in td.items():
                broads.append([rl[(nid(_f), nid(_t))], [lane_id[id(l)] for l in ls]])
        respawn = [[nid(r.start_node), nid(r.end_node)] for r in b.get_respawn_roads()]
        trig = b.pre_block_socket.positive_road
        blocks.append(
            dict(
                id=b.ID, sockets=sockets, spawn_lanes=spawn_lanes, respawn_roads=respawn, roads=broads,
                trigger_road=[nid(trig.start_node), nid(trig.end_node)] if b.block_index != 0 else None,
                config={k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in b.get_config().items()}
                if b.block_index != 0 else {}, pre_socket=b.pre_block_socket_index,
            )
        )
    return dict(
        seed=seed, lane_num=lane_num, lane_width=lane_width, exit_length=exit_length, nodes=nodes, roads=roads,
        lanes=lanes, blocks=blocks, boxes=np.array(boxes, dtype=np.float64).reshape(-1, 7), net=net, big=big,
    )
