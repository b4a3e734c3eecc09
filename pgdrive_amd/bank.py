"""Map-description bank I/O (BaseMap.save_map / MapManager.read_all_maps analogue: base_map.py:103-130,
map_manager.py:43-91).  The shipped bank `assets/pg_bank_v0.json.gz` holds the PGDrive-v0 seed range 1000..1099."""
import gzip
import json
import os

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
DEFAULT_BANK = os.path.join(_ASSETS, "pg_bank_v0.json.gz")
MA_ROUNDABOUT_BANK = os.path.join(_ASSETS, "ma_roundabout_v0.json.gz")  # MARoundaboutMap (marl_inout_roundabout.py:30-63)


def load_descriptions(path=DEFAULT_BANK):
    with gzip.open(path, "rb") as f:
        data = json.loads(f.read().decode())
    return data["maps"]


def save_descriptions(descs, path, source="pgdrive_amd"):
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(dict(version=0, source=source, maps=descs), separators=(",", ":")).encode())
