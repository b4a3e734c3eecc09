"""Map-description bank I/O (BaseMap.save_map / MapManager.read_all_maps analogue: base_map.py:103-130,
map_manager.py:43-91).  The shipped bank `assets/pg_bank_v0.json.gz` holds the PGDrive-v0 seed range 1000..1099."""
import gzip
import json
import os

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
DEFAULT_BANK = os.path.join(_ASSETS, "pg_bank_v0.json.gz")
MA_ROUNDABOUT_BANK = os.path.join(_ASSETS, "ma_roundabout_v0.json.gz")  # MARoundaboutMap (marl_inout_roundabout.py:30-63)


def load_descriptions(path=DEFAULT_BANK):
    """Reads either our flattened description bank (.json.gz, key "maps") or a map file in the REFERENCE's format
    (PGDriveEnv.dump_all_maps: {"map_config", "map_data": {seed: {"block_sequence": [...]}}}, plain or gzipped JSON); the
    latter is rebuilt block by block from the saved parameters (pgdrive_amd/mapgen.py:generate_from_block_sequence)."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    data = json.loads(raw.decode())
    if "map_data" in data:
        from . import mapgen
        return mapgen.load_all_maps(data)
    return data["maps"]


def save_descriptions(descs, path, source="pgdrive_amd"):
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(dict(version=0, source=source, maps=descs), separators=(",", ":")).encode())


_CACHE = {}


def randomized_lane_config(seed, lane_num, lane_width, random_lane_width=False, random_lane_num=False):
    """MapManager.add_random_to_map (manager/map_manager.py:157-169): the manager's RandomState is re-seeded with the map
    seed at every reset (base_engine.py:300-304), so the draws are a function of the seed: width uniform in
    [MIN_LANE_WIDTH, MAX_LANE_WIDTH) = [3.0, 4.5), then lane count randint(MIN_LANE_NUM, MAX_LANE_NUM) = randint(2, 3) -- the
    upper bound is exclusive, i.e. always 2 (pg_map.py:13-16), reproduced as is."""
    from .scenario import get_np_random
    rs = get_np_random(seed)
    if random_lane_width:
        lane_width = rs.rand() * (4.5 - 3.0) + 3.0
    if random_lane_num:
        lane_num = int(rs.randint(2, 3))
    return lane_num, lane_width


def get_descriptions(seeds, lane_num=3, lane_width=3.5, exit_length=50, block_num=3, block_seq=None,
                     random_lane_width=False, random_lane_num=False):
    """Map descriptions for `seeds`, generated on the host by our own BIG (pgdrive_amd/mapgen.py) and cached per process
    (MapManager's per-seed PGMap cache, manager/map_manager.py:98-155)."""
    from . import mapgen
    out = []
    for s in seeds:
        ln, lw = lane_num, lane_width
        if random_lane_width or random_lane_num:
            ln, lw = randomized_lane_config(int(s), lane_num, lane_width, random_lane_width, random_lane_num)
        key = (int(s), ln, lw, exit_length, block_num, block_seq)
        if key not in _CACHE:
            _CACHE[key] = mapgen.generate_map(int(s), ln, lw, exit_length, block_num, block_seq)
        out.append(_CACHE[key])
    return out
