"""Generate the map-description bank with the REFERENCE's BIG generator (runs only in the build container).

    PYTHONHASHSEED=0 python oracle/gen_mapbank.py

Output: pgdrive_amd/assets/pg_bank_v0.json.gz — data only (lane geometry, roads, block metadata) for the
`PGDrive-v0` seed range 1000..1099 (pgdrive/register.py:14-17; map=3, lane_num=3, lane_width=3.5, exit_length=50,
pgdrive/envs/pgdrive_env.py:31-37).  Also writes tests/golden/boxes_seed*.npz: the boxes the reference's block code
registered in (stubbed) Bullet for a few seeds — the golden for pgdrive_amd.mapdata.build_boxes.
"""
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_export  # noqa: E402


def strip(m):
    return {k: v for k, v in m.items() if k not in ("net", "big", "boxes")}


def main():
    root = os.path.dirname(HERE)
    descs = []
    for seed in range(1000, 1100):
        m = ref_export.generate(seed, block_num=3)
        descs.append(strip(m))
        if seed in (1000, 1003, 1017, 1042, 1099):
            np.savez_compressed(os.path.join(root, "tests", "golden", "boxes_seed%d.npz" % seed), boxes=m["boxes"])
    out = os.path.join(root, "pgdrive_amd", "assets", "pg_bank_v0.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps(dict(version=0, source="decisionforce/pgdrive v0.1.4 BIG, seeds 1000..1099, map=3",
                                maps=descs), separators=(",", ":")).encode())
    print("wrote", out, os.path.getsize(out))


def main_marl():
    """MARoundaboutMap (envs/marl_envs/marl_inout_roundabout.py:30-63) -> pgdrive_amd/assets/ma_roundabout_v0.json.gz
    + tests/golden/boxes_ma_roundabout.npz"""
    root = os.path.dirname(HERE)
    m = ref_export.generate_ma_roundabout()
    np.savez_compressed(os.path.join(root, "tests", "golden", "boxes_ma_roundabout.npz"), boxes=m["boxes"])
    out = os.path.join(root, "pgdrive_amd", "assets", "ma_roundabout_v0.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps(dict(version=0, source="decisionforce/pgdrive v0.1.4 MARoundaboutMap (lane_num=2, "
                                "exit_length=60, exit_radius=10, inner_radius=30, angle=70)", maps=[strip(m)]),
                           separators=(",", ":")).encode())
    print("wrote", out, os.path.getsize(out))
    # MAIntersectionMap (envs/marl_envs/marl_intersection.py:29-54) -> tests/golden/ma_intersection_v0.json.gz (+ boxes)
    m = ref_export.generate_ma_intersection()
    np.savez_compressed(os.path.join(root, "tests", "golden", "boxes_ma_intersection.npz"), boxes=m["boxes"])
    out = os.path.join(root, "tests", "golden", "ma_intersection_v0.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps(dict(version=0, source="decisionforce/pgdrive v0.1.4 MAIntersectionMap (lane_num=2, "
                                "exit_length=60, u-turns)", maps=[strip(m)]), separators=(",", ":")).encode())
    print("wrote", out, os.path.getsize(out))


def main_marl_bottleneck():
    """MABottleneckMap (envs/marl_envs/marl_bottleneck.py:28-67) -> tests/golden/ma_bottleneck_v0.json.gz (+ boxes)"""
    root = os.path.dirname(HERE)
    m = ref_export.generate_ma_bottleneck()
    np.savez_compressed(os.path.join(root, "tests", "golden", "boxes_ma_bottleneck.npz"), boxes=m["boxes"])
    out = os.path.join(root, "tests", "golden", "ma_bottleneck_v0.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps(dict(version=0, source="decisionforce/pgdrive v0.1.4 MABottleneckMap (4 lanes, neck 1 lane x 20 m)",
                                maps=[strip(m)]), separators=(",", ":")).encode())
    print("wrote", out, os.path.getsize(out))


def main_mapjson():
    """BaseMap.save_map (component/map/base_map.py:103-118) of BIG-generated maps -> tests/golden/mapjson_v0.json: the
    reference's own block_sequence JSON (block parameters, id, pre_block_socket_index) per seed."""
    root = os.path.dirname(HERE)
    out = {}
    for seed, kw in [(1000, dict(block_num=3)), (1003, dict(block_num=3)), (1042, dict(block_num=3)), (0, dict(block_num=7)),
                     (10, dict(block_seq="CrXRTOS")), (11, dict(block_num=4, lane_num=2))]:
        m = ref_export.generate(seed, **kw)
        seq = []
        for b in m["big"].blocks:  # == BaseMap.save_map
            cfg = b.get_config().get_serializable_dict()
            cfg["id"] = b.ID
            cfg["pre_block_socket_index"] = b.pre_block_socket_index
            seq.append(cfg)
        out[str(seed)] = dict(kw=kw, block_sequence=seq)
    with open(os.path.join(root, "tests", "golden", "mapjson_v0.json"), "w") as f:
        json.dump(out, f)
    print("wrote mapjson goldens", {k: [b["id"] for b in v["block_sequence"]] for k, v in out.items()})


def main_marl_parking_lot():
    """MAParkingLotMap (envs/marl_envs/marl_parking_lot.py:92-130) -> tests/golden/ma_parking_lot_v0.json.gz (+ boxes)"""
    root = os.path.dirname(HERE)
    m = ref_export.generate_ma_parking_lot()
    np.savez_compressed(os.path.join(root, "tests", "golden", "boxes_ma_parking_lot.npz"), boxes=m["boxes"])
    d = strip(m)
    out = os.path.join(root, "tests", "golden", "ma_parking_lot_v0.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps(dict(version=0, source="decisionforce/pgdrive v0.1.4 MAParkingLotMap (8 spaces)", maps=[d]),
                           separators=(",", ":")).encode())
    print("wrote", out, os.path.getsize(out), "spaces", len(m["parking_space"]))


def main_marl_tollgate():
    """MATollGateMap (envs/marl_envs/marl_tollgate.py:108-160) -> tests/golden/ma_tollgate_v0.json.gz (+ boxes, booths)"""
    root = os.path.dirname(HERE)
    m = ref_export.generate_ma_tollgate()
    np.savez_compressed(os.path.join(root, "tests", "golden", "boxes_ma_tollgate.npz"), boxes=m["boxes"])
    d = strip(m)
    d["booths"] = m["booths"]
    out = os.path.join(root, "tests", "golden", "ma_tollgate_v0.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps(dict(version=0, source="decisionforce/pgdrive v0.1.4 MATollGateMap (3 -> 8 -> 3 lanes)", maps=[d]),
                           separators=(",", ":")).encode())
    print("wrote", out, os.path.getsize(out), "booths", len(m["booths"]))


def main_mapgen_goldens():
    """Extra goldens for pgdrive_amd/mapgen.py beyond the 100-seed bank: other block counts, lane counts / widths and
    explicit block sequences through every block type -> tests/golden/mapgen_v0.json.gz"""
    root = os.path.dirname(HERE)
    cases = []
    for seed, kw in [(0, dict(block_num=7)), (1, dict(block_num=7)), (2, dict(block_num=5)), (3, dict(block_num=1)),
                     (11, dict(block_num=4, lane_num=2)), (12, dict(block_num=4, lane_num=2, lane_width=3.0)),
                     (13, dict(block_num=3, lane_num=3, lane_width=4.5, exit_length=60)),
                     (5, dict(block_seq="SCS")), (6, dict(block_seq="XTO")), (7, dict(block_seq="rRC")),
                     (8, dict(block_seq="OOO")), (9, dict(block_seq="TTT")), (10, dict(block_seq="CrXRTOS")),
                     (4242, dict(block_num=10))]:
        m = ref_export.generate(seed, **kw)
        cases.append(dict(seed=seed, kw=kw, desc=strip(m)))
    out = os.path.join(root, "tests", "golden", "mapgen_v0.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps(dict(cases=cases), separators=(",", ":")).encode())
    print("wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    if "--parking" in sys.argv:
        main_marl_parking_lot()
    elif "--tollgate" in sys.argv:
        main_marl_tollgate()
    elif "--mapjson" in sys.argv:
        main_mapjson()
    elif "--bottleneck" in sys.argv:
        main_marl_bottleneck()
    elif "--marl" in sys.argv:
        main_marl()
    elif "--mapgen" in sys.argv:
        main_mapgen_goldens()
    else:
        main()
