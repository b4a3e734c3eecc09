"""Host-side flattening of a PG map description into the immutable device tables of `include/pgdrive_hip.h`.

A *map description* is a plain dict (lanes / roads / nodes / blocks) — produced either by our own block-incremental
generator or read from the map bank.  This module derives everything the step engine needs from it:

* lane records   (pgdrive/component/lane/straight_lane.py:13-67, circular_lane.py:11-67)
* successor table (AbstractLane.is_previous_lane_of, abs_lane.py:114-119)
* the boxes the reference hands to Bullet: lane-surface boxes, lane-line ghost boxes, sidewalk bodies
  (pgdrive/component/blocks/base_block.py:180-464, constants.py:226-251)
* a uniform grid over the boxes (replaces Bullet's broadphase for rayTestAll/contactTest/sweep queries,
  utils/scene_utils.py:138-231, base_vehicle.py:615-644)
* BFS routes (RoadNetwork.shortest_path, road_network.py:241-269)

Everything here runs once per map on the host; the result is uploaded with `pgd_upload_maps`.
"""
import math

import numpy as np

# DrivableAreaProperty (constants.py:226-251)
CIRCULAR_SEGMENT_LENGTH = 4.0
STRIPE_LENGTH = 1.5
LANE_LINE_WIDTH = 0.15
SIDEWALK_LENGTH = 3.0
SIDEWALK_WIDTH = 3.0
SIDEWALK_LINE_DIST = 0.6

LT_NONE, LT_BROKEN, LT_CONTINUOUS, LT_SIDE = 0, 1, 2, 3
BOX_LANE, BOX_WHITE, BOX_YELLOW, BOX_BROKEN, BOX_SIDEWALK = 0, 1, 2, 3, 4

MAX_SUCC = 8
MAX_CKPT = 32

LANE_DT = np.dtype(
    [
        ("ax", "<f4"), ("ay", "<f4"), ("bx", "<f4"), ("by", "<f4"), ("c", "<f4"), ("dir", "<f4"), ("length", "<f4"),
        ("width", "<f4"), ("ex", "<f4"), ("ey", "<f4"), ("road", "<i2"), ("index", "<i2"), ("n_succ", "<i2"),
        ("pad", "<i2"), ("succ", "<i2", (MAX_SUCC, ))
    ]
)
ROAD_DT = np.dtype(
    [
        ("frm", "<i2"), ("to", "<i2"), ("first_lane", "<i2"), ("n_lanes", "<i2"), ("negative", "u1"),
        ("block_id", "u1"), ("valid", "u1"), ("pad0", "u1"), ("pad1", "<i4")
    ]
)
BOX_DT = np.dtype(
    [("cx", "<f4"), ("cy", "<f4"), ("ux", "<f4"), ("uy", "<f4"), ("hl", "<f4"), ("hw", "<f4"), ("kind", "<i4"),
     ("lane", "<i4")]
)
MAP_DT = np.dtype(
    [
        ("lane_off", "<i4"), ("n_lanes", "<i4"), ("road_off", "<i4"), ("n_roads", "<i4"), ("box_off", "<i4"),
        ("n_boxes", "<i4"), ("cell_off", "<i4"), ("item_off", "<i4"), ("gx", "<i4"), ("gy", "<i4"), ("ox", "<f4"),
        ("oy", "<f4"), ("cell", "<f4"), ("lane_width", "<f4"), ("pad", "<i4", (2, ))
    ]
)
assert LANE_DT.itemsize == 64 and ROAD_DT.itemsize == 16 and BOX_DT.itemsize == 32 and MAP_DT.itemsize == 64


# ----------------------------------------------------------------------------------------------------------------------
# lane closed forms on description dicts (float64, host only)
# ----------------------------------------------------------------------------------------------------------------------
def lane_position(l, longitudinal, lateral):
    """StraightLane.position (straight_lane.py:53-54) / CircularLane.position (circular_lane.py:41-44)."""
    if l["type"] == 0:
        dx, dy = l["direction"]
        return (
            l["start"][0] + longitudinal * dx + lateral * -dy,
            l["start"][1] + longitudinal * dy + lateral * dx,
        )
    d = l["direction"]
    phi = d * longitudinal / l["radius"] + l["start_phase"]
    r = l["radius"] - lateral * d
    return (l["center"][0] + r * math.cos(phi), l["center"][1] + r * math.sin(phi))


def lane_heading_at(l, longitudinal):
    if l["type"] == 0:
        return l["heading"]
    d = l["direction"]
    phi = d * longitudinal / l["radius"] + l["start_phase"]
    return phi + math.pi / 2 * d


def wrap_to_pi(x):
    return ((x + math.pi) % (2 * math.pi)) - math.pi


def lane_local_coordinates(l, pos):
    """straight_lane.py:62-67 / circular_lane.py:57-67."""
    if l["type"] == 0:
        dx, dy = l["direction"]
        ddx, ddy = pos[0] - l["start"][0], pos[1] - l["start"][1]
        return ddx * dx + ddy * dy, ddx * -dy + ddy * dx
    ddx, ddy = pos[0] - l["center"][0], pos[1] - l["center"][1]
    phi = math.atan2(ddy, ddx)
    phi = l["start_phase"] + wrap_to_pi(phi - l["start_phase"])
    r = math.sqrt(ddx * ddx + ddy * ddy)
    d = l["direction"]
    return d * (phi - l["start_phase"]) * l["radius"], d * (l["radius"] - r)


# ----------------------------------------------------------------------------------------------------------------------
# boxes (what the reference registers in Bullet)
# ----------------------------------------------------------------------------------------------------------------------
def _line_kind(line_type, color):
    """base_block.py:293-296: prohibit -> white/yellow continuous by colour, else broken."""
    if line_type in (LT_CONTINUOUS, LT_SIDE):
        return BOX_WHITE if color == 0 else BOX_YELLOW
    return BOX_BROKEN


def _seg_box(out, kind, start, end, middle, half_len, half_w, lane=-1):
    theta = math.atan2(end[1] - start[1], end[0] - start[0])
    out.append((kind, middle[0], middle[1], theta, half_len, half_w, lane))


def _add_lane_line(out, start, end, middle, color, line_type, straight_stripe=False):
    """_add_lane_line2bullet (base_block.py:316-365): ghost box 0.15 m wide; BROKEN stripes get `length` as half extent."""
    length = math.hypot(end[0] - start[0], end[1] - start[1])
    if length <= 0 or straight_stripe:
        return
    hl = length / 2 if line_type != LT_BROKEN else length
    _seg_box(out, _line_kind(line_type, color), start, end, middle, hl, LANE_LINE_WIDTH / 2)


def _add_sidewalk(out, start, end, middle, radius, direction):
    """_add_sidewalk2bullet (base_block.py:367-394): unit box scaled to (length*factor, 3 m), pushed 2.1 m outwards."""
    length = math.hypot(end[0] - start[0], end[1] - start[1])
    if radius == 0:
        factor = 1.0
    elif direction == 1:
        factor = 1 - SIDEWALK_LINE_DIST / radius
    else:
        factor = (1 + SIDEWALK_WIDTH / radius) * (1 + SIDEWALK_LINE_DIST / radius)
    dvx, dvy = end[0] - start[0], end[1] - start[1]
    n = math.hypot(dvx, dvy)
    vx, vy = -dvy / n, dvx / n
    off = SIDEWALK_WIDTH / 2 + SIDEWALK_LINE_DIST
    mid = (middle[0] + vx * off, middle[1] + vy * off)
    theta = math.atan2(dvy, dvx)
    out.append((BOX_SIDEWALK, mid[0], mid[1], theta, 0.5 * length * factor, 0.5 * SIDEWALK_WIDTH, -1))


def _mid(a, b):
    return ((a[0] + b[0]) / 2, (a[1] + b[1]) / 2)


def _add_lane_lines(out, l, lane_idx):
    """_add_pgdrive_lanes (base_block.py:180-265)."""
    w = l["width"]
    straight = l["type"] == 0
    for k, i in enumerate((-1, 1)):
        lt = l["line_types"][k]
        color = l["line_colors"][k]
        if lt == LT_NONE or (lane_idx != 0 and k == 0):
            if straight:
                continue
            elif l["radius"] != w / 2:
                continue
        lat = i * w / 2
        if lt in (LT_CONTINUOUS, LT_SIDE):
            if straight:
                s, e = lane_position(l, 0, lat), lane_position(l, l["length"], lat)
                _add_lane_line(out, s, e, lane_position(l, l["length"] / 2, lat), color, lt)
            else:
                n = int(l["length"] / CIRCULAR_SEGMENT_LENGTH)
                for seg in range(n):
                    s = lane_position(l, seg * CIRCULAR_SEGMENT_LENGTH, lat)
                    e = lane_position(l, (seg + 1) * CIRCULAR_SEGMENT_LENGTH, lat)
                    _add_lane_line(out, s, e, _mid(s, e), color, lt)
                s = lane_position(l, n * CIRCULAR_SEGMENT_LENGTH, lat)
                e = lane_position(l, l["length"], lat)
                _add_lane_line(out, s, e, _mid(s, e), color, lt)
            if lt == LT_SIDE:
                radius = l["radius"] if not straight else 0.0
                direction = l["direction"] if not straight else 0
                n = int(l["length"] / SIDEWALK_LENGTH)
                for seg in range(n):
                    s = lane_position(l, seg * SIDEWALK_LENGTH, lat)
                    e = lane_position(l, (seg + 1) * SIDEWALK_LENGTH, lat)
                    _add_sidewalk(out, s, e, _mid(s, e), radius, direction)
                s = lane_position(l, n * SIDEWALK_LENGTH, lat)
                e = lane_position(l, l["length"], lat)
                if math.hypot(s[0] - e[0], s[1] - e[1]) > 1e-1:
                    _add_sidewalk(out, s, e, _mid(s, e), radius, direction)
        elif lt == LT_BROKEN:
            n = int(l["length"] / (2 * STRIPE_LENGTH))
            for seg in range(n):
                s = lane_position(l, seg * STRIPE_LENGTH * 2, lat)
                e = lane_position(l, seg * STRIPE_LENGTH * 2 + STRIPE_LENGTH, lat)
                m = lane_position(l, seg * STRIPE_LENGTH * 2 + STRIPE_LENGTH / 2, lat)
                _add_lane_line(out, s, e, m, color, lt, straight)
            s = lane_position(l, n * STRIPE_LENGTH * 2, lat)
            e = lane_position(l, l["length"] + STRIPE_LENGTH, lat)
            if not straight:
                _add_lane_line(out, s, e, _mid(s, e), color, lt, straight)
            else:
                # one ghost box over the whole straight lane (_add_box_body, base_block.py:285-314)
                s, e = lane_position(l, 0, lat), lane_position(l, l["length"], lat)
                length = math.hypot(e[0] - s[0], e[1] - s[1])
                _seg_box(out, _line_kind(lt, color), s, e, lane_position(l, l["length"] / 2, lat), length / 2,
                         LANE_LINE_WIDTH / 2)


def _add_lane_surface(out, lanes, lane_ids):
    """_add_lane_surface/_add_lane2bullet (base_block.py:396-456): (len+0.1) x (width+1.2); arcs chopped into chords."""
    if lanes[0]["type"] == 0:
        for l, lid in zip(lanes, lane_ids):
            mid = lane_position(l, l["length"] / 2, 0)
            end = lane_position(l, l["length"], 0)
            theta = math.atan2(end[1] - mid[1], end[0] - mid[0])
            out.append((BOX_LANE, mid[0], mid[1], theta, (l["length"] + 0.1) / 2, (l["width"] + 2 * SIDEWALK_LINE_DIST) / 2,
                        lid))
    else:
        for l, lid in zip(lanes, lane_ids):
            n = int(l["length"] / CIRCULAR_SEGMENT_LENGTH)
            for i in range(n):
                mid = lane_position(l, l["length"] * (i + .5) / n, 0)
                end = lane_position(l, l["length"] * (i + 1) / n, 0)
                theta = math.atan2(end[1] - mid[1], end[0] - mid[0])
                out.append(
                    (BOX_LANE, mid[0], mid[1], theta, (l["length"] * 1.3 / n + 0.1) / 2,
                     (l["width"] + 2 * SIDEWALK_LINE_DIST) / 2, lid)
                )


def build_boxes(desc):
    """All boxes of a map in the reference's creation order: block by block, road by road
    (BaseBlock._create_in_world, base_block.py:142-156).  Returns float64 [n,7]: kind,cx,cy,theta,hl,hw,lane."""
    out = []
    lanes = desc["lanes"]
    for b in desc["blocks"]:
        for _road, lane_ids in b["roads"]:
            ls = [lanes[i] for i in lane_ids]
            if not ls:
                continue
            _add_lane_surface(out, ls, lane_ids)
            for idx, (l, lid) in enumerate(zip(ls, lane_ids)):
                _add_lane_lines(out, l, idx)
    return np.array(out, dtype=np.float64).reshape(-1, 7)


# ----------------------------------------------------------------------------------------------------------------------
# topology helpers
# ----------------------------------------------------------------------------------------------------------------------
def build_successors(desc):
    """succ[a] = [b : |a.end - b.start| < 0.1] (AbstractLane.is_previous_lane_of, abs_lane.py:114-119)."""
    lanes = desc["lanes"]
    starts = np.array([l["start"] for l in lanes], dtype=np.float64)
    ends = np.array([l["end"] for l in lanes], dtype=np.float64)
    succ = []
    for a in range(len(lanes)):
        d = np.hypot(starts[:, 0] - ends[a, 0], starts[:, 1] - ends[a, 1])
        succ.append([int(b) for b in np.nonzero(d < 1e-1)[0]])
    return succ


def graph_of(desc):
    """node -> ordered list of (to_node, road_id) in RoadNetwork.graph insertion order."""
    g = {}
    for rid, r in enumerate(desc["roads"]):
        g.setdefault(r["frm"], []).append((r["to"], rid))
    return g


def shortest_path(desc, start, goal):
    """RoadNetwork.bfs_paths/shortest_path (road_network.py:241-269), on node ids.

    The reference iterates `set(next_nodes) - set(path)`, i.e. in string-hash order; ties between equal-length routes
    are therefore process-dependent there.  We iterate in graph insertion order (deterministic)."""
    assert start != goal
    g = graph_of(desc)
    queue = [(start, [start])]
    while queue:
        node, path = queue.pop(0)
        if node not in g:
            return []
        for nxt, _ in g[node]:
            if nxt in path:
                continue
            if nxt == goal:
                return path + [nxt]
            elif nxt in g:
                queue.append((nxt, path + [nxt]))
    return []


def road_lookup(desc):
    return {(r["frm"], r["to"]): rid for rid, r in enumerate(desc["roads"])}


# ----------------------------------------------------------------------------------------------------------------------
# uniform grid over boxes
# ----------------------------------------------------------------------------------------------------------------------
def box_aabb(boxes):
    c, s = np.cos(boxes[:, 3]), np.sin(boxes[:, 3])
    ex = np.abs(c) * boxes[:, 4] + np.abs(s) * boxes[:, 5]
    ey = np.abs(s) * boxes[:, 4] + np.abs(c) * boxes[:, 5]
    return boxes[:, 1] - ex, boxes[:, 1] + ex, boxes[:, 2] - ey, boxes[:, 2] + ey


def build_grid(boxes, cell=8.0, margin=0.05):
    """CSR grid: cell -> ascending box ids whose (slightly inflated) AABB touches the cell."""
    x0, x1, y0, y1 = box_aabb(boxes)
    ox = math.floor((x0.min() - 1.0) / cell) * cell
    oy = math.floor((y0.min() - 1.0) / cell) * cell
    gx = int(math.floor((x1.max() + 1.0 - ox) / cell)) + 1
    gy = int(math.floor((y1.max() + 1.0 - oy) / cell)) + 1
    cx0 = np.clip(np.floor((x0 - margin - ox) / cell).astype(int), 0, gx - 1)
    cx1 = np.clip(np.floor((x1 + margin - ox) / cell).astype(int), 0, gx - 1)
    cy0 = np.clip(np.floor((y0 - margin - oy) / cell).astype(int), 0, gy - 1)
    cy1 = np.clip(np.floor((y1 + margin - oy) / cell).astype(int), 0, gy - 1)
    lists = [[] for _ in range(gx * gy)]
    for b in range(len(boxes)):
        for cy in range(cy0[b], cy1[b] + 1):
            for cx in range(cx0[b], cx1[b] + 1):
                lists[cy * gx + cx].append(b)
    start = np.zeros(gx * gy + 1, dtype=np.int32)
    start[1:] = np.cumsum([len(c) for c in lists])
    items = np.array([b for c in lists for b in c], dtype=np.int32)
    return dict(ox=ox, oy=oy, cell=cell, gx=gx, gy=gy, start=start, items=items)


# ----------------------------------------------------------------------------------------------------------------------
# packing
# ----------------------------------------------------------------------------------------------------------------------
def pack_lanes(desc, succ, truncate_succ=False):
    n = len(desc["lanes"])
    out = np.zeros(n, dtype=LANE_DT)
    for i, l in enumerate(desc["lanes"]):
        r = out[i]
        if l["type"] == 0:
            r["ax"], r["ay"] = l["start"]
            r["bx"], r["by"] = l["direction"]
            r["c"] = l["heading"]
            r["dir"] = 0.0
        else:
            r["ax"], r["ay"] = l["center"]
            r["bx"], r["by"] = l["radius"], l["start_phase"]
            r["c"] = l["end_phase"]
            r["dir"] = float(l["direction"])
        r["length"], r["width"] = l["length"], l["width"]
        r["ex"], r["ey"] = l["end"]
        r["road"], r["index"] = l["road"], l["index"]
        s = succ[i]
        if len(s) > MAX_SUCC:
            # the successor list only feeds the IDM traffic (neighbour search, routing): maps without IDM traffic (the
            # multi-agent parking lot: 9 ways out of one lane) may drop the surplus
            if not truncate_succ:
                raise ValueError("lane %d has %d successors > PGD_MAX_SUCC" % (i, len(s)))
            s = s[:MAX_SUCC]
        r["n_succ"] = len(s)
        r["succ"][:] = -1
        r["succ"][:len(s)] = s
    return out


def pack_lanes_f64(desc):
    """The ten float fields of every lane record in float64, same choice of parameters as pack_lanes (test infrastructure: the
    CPU oracle's float64 table path, oracle/pgd_oracle.c::orc_upload_tables_f64)."""
    out = np.zeros((len(desc["lanes"]), 10), dtype=np.float64)
    for i, l in enumerate(desc["lanes"]):
        if l["type"] == 0:
            out[i, :6] = [l["start"][0], l["start"][1], l["direction"][0], l["direction"][1], l["heading"], 0.0]
        else:
            out[i, :6] = [l["center"][0], l["center"][1], l["radius"], l["start_phase"], l["end_phase"], float(l["direction"])]
        out[i, 6:] = [l["length"], l["width"], l["end"][0], l["end"][1]]
    return out


def pack_roads(desc):
    out = np.zeros(len(desc["roads"]), dtype=ROAD_DT)
    for i, r in enumerate(desc["roads"]):
        out[i]["frm"], out[i]["to"] = r["frm"], r["to"]
        out[i]["first_lane"], out[i]["n_lanes"] = r["first_lane"], r["n_lanes"]
        out[i]["negative"] = 1 if r["negative"] else 0
        out[i]["block_id"] = ord(r["block_id"][0])
        out[i]["valid"] = 1 if r["valid"] else 0
    return out


def pack_boxes(boxes):
    out = np.zeros(len(boxes), dtype=BOX_DT)
    out["kind"] = boxes[:, 0].astype(np.int32)
    out["cx"], out["cy"] = boxes[:, 1], boxes[:, 2]
    out["ux"], out["uy"] = np.cos(boxes[:, 3]), np.sin(boxes[:, 3])
    out["hl"], out["hw"] = boxes[:, 4], boxes[:, 5]
    out["lane"] = boxes[:, 6].astype(np.int32)
    return out


class MapBank:
    """Concatenated device tables for a list of map descriptions."""
    def __init__(self, descs, cell=8.0, truncate_succ=False):
        self.descs = list(descs)
        maps = np.zeros(len(self.descs), dtype=MAP_DT)
        lanes, roads, boxes, cstart, citems = [], [], [], [], []
        lanes64, boxes64, maps64 = [], [], []  # the same float fields before their rounding to float32 (oracle pinning only)
        lo = ro = bo = co = io = 0
        self.succ = []
        for m, d in enumerate(self.descs):
            succ = build_successors(d)
            self.succ.append(succ)
            L = pack_lanes(d, succ, truncate_succ)
            R = pack_roads(d)
            bx = build_boxes(d)
            B = pack_boxes(bx)
            g = build_grid(bx, cell)
            h = maps[m]
            h["lane_off"], h["n_lanes"] = lo, len(L)
            h["road_off"], h["n_roads"] = ro, len(R)
            h["box_off"], h["n_boxes"] = bo, len(B)
            h["cell_off"], h["item_off"] = co, io
            h["gx"], h["gy"], h["ox"], h["oy"], h["cell"] = g["gx"], g["gy"], g["ox"], g["oy"], g["cell"]
            h["lane_width"] = d["lane_width"]
            lanes.append(L), roads.append(R), boxes.append(B), cstart.append(g["start"]), citems.append(g["items"])
            lanes64.append(pack_lanes_f64(d))
            boxes64.append(np.stack([bx[:, 1], bx[:, 2], np.cos(bx[:, 3]), np.sin(bx[:, 3]), bx[:, 4], bx[:, 5]], axis=1)
                           if len(bx) else np.zeros((0, 6)))
            maps64.append([g["ox"], g["oy"], g["cell"], d["lane_width"]])
            lo += len(L)
            ro += len(R)
            bo += len(B)
            co += len(g["start"])
            io += len(g["items"])
        self.maps = maps
        self.lanes = np.concatenate(lanes)
        self.roads = np.concatenate(roads)
        self.boxes = np.concatenate(boxes)
        self.cell_start = np.concatenate(cstart).astype(np.int32)
        self.cell_items = np.concatenate(citems).astype(np.int32)
        self.lanes64 = np.ascontiguousarray(np.concatenate(lanes64), dtype=np.float64)
        self.boxes64 = np.ascontiguousarray(np.concatenate(boxes64), dtype=np.float64)
        self.maps64 = np.ascontiguousarray(np.array(maps64, dtype=np.float64).reshape(-1, 4))

    def nbytes(self):
        return sum(a.nbytes for a in (self.maps, self.lanes, self.roads, self.boxes, self.cell_start, self.cell_items))
