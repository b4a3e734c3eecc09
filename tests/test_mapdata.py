"""Host-side geometry: our block->box flattening reproduces, box for box, what the reference's own block code registered
in Bullet (golden: tests/golden/boxes_seed*.npz recorded by oracle/gen_mapbank.py through the import stubs)."""
import os

import numpy as np
import pytest

from pgdrive_amd import mapdata

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("seed", [1000, 1003, 1017, 1042, 1099])
def test_boxes_match_reference_recording(descs, seed):
    d = [m for m in descs if m["seed"] == seed][0]
    ref = np.load(os.path.join(GOLD, "boxes_seed%d.npz" % seed))["boxes"]
    mine = mapdata.build_boxes(d)
    assert ref.shape == mine.shape
    assert (ref[:, 0] == mine[:, 0]).all() and (ref[:, 6] == mine[:, 6]).all()  # kind, lane id, creation order
    assert np.abs(ref[:, [1, 2, 4, 5]] - mine[:, [1, 2, 4, 5]]).max() < 1e-9
    assert np.abs(np.sin(ref[:, 3] - mine[:, 3])).max() < 1e-9


def test_box_census(descs):
    """42-138 lanes, 280-420 lane boxes ... per map; every kind present."""
    for d in descs[:10]:
        bx = mapdata.build_boxes(d)
        kinds = bx[:, 0].astype(int)
        assert 20 <= len(d["lanes"]) <= 200
        assert (kinds == 0).sum() >= len(d["lanes"]) - 8  # arcs shorter than 4 m get no surface box
        for k in (1, 2, 3, 4):
            assert (kinds == k).sum() > 0


def test_grid_covers_every_box(descs):
    """Each box is listed in every cell its AABB touches, lists are ascending (= Bullet insertion order)."""
    d = descs[3]
    bx = mapdata.build_boxes(d)
    g = mapdata.build_grid(bx, cell=8.0)
    rng = np.random.default_rng(0)
    for b in rng.integers(0, len(bx), 200):
        _, cx, cy, th, hl, hw, _ = bx[b]
        for _ in range(4):
            a, c = rng.uniform(-hl, hl), rng.uniform(-hw, hw)
            px, py = cx + a * np.cos(th) - c * np.sin(th), cy + a * np.sin(th) + c * np.cos(th)
            ix, iy = int((px - g["ox"]) // g["cell"]), int((py - g["oy"]) // g["cell"])
            cell = iy * g["gx"] + ix
            items = g["items"][g["start"][cell]:g["start"][cell + 1]]
            assert b in items
            assert (np.diff(items) > 0).all()


def test_successors_and_routes(descs):
    d = descs[0]
    succ = mapdata.build_successors(d)
    assert max(len(s) for s in succ) <= mapdata.MAX_SUCC
    # the spawn road '>' -> '>>' leads into '>>' -> '>>>' lane by lane (first_block.py:44-74)
    n = d["nodes"]
    rl = mapdata.road_lookup(d)
    r0, r1 = d["roads"][rl[(n.index(">"), n.index(">>"))]], d["roads"][rl[(n.index(">>"), n.index(">>>"))]]
    for k in range(r0["n_lanes"]):
        assert r1["first_lane"] + k in succ[r0["first_lane"] + k]
    path = mapdata.shortest_path(d, n.index(">"), n.index(">>>"))
    assert [n[i] for i in path] == [">", ">>", ">>>"]


def test_bank_packing_roundtrip(descs):
    mb = mapdata.MapBank(descs[:3])
    assert len(mb.maps) == 3
    for m, d in zip(mb.maps, descs[:3]):
        L = mb.lanes[m["lane_off"]:m["lane_off"] + m["n_lanes"]]
        assert len(L) == len(d["lanes"])
        for rec, l in zip(L, d["lanes"]):
            assert abs(rec["length"] - l["length"]) < 1e-4 and rec["road"] == l["road"] and rec["index"] == l["index"]
        assert m["n_boxes"] > 0 and m["gx"] * m["gy"] > 0
    assert mb.nbytes() < 4 << 20
