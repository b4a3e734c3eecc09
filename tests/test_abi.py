"""The C-ABI library loads, exports every symbol include/pgdrive_hip.h declares, and the Python mirrors of the structs
have the C sizes (no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np

from pgdrive_amd import _abi, build, mapdata, scenario

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "pgdrive_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pgd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from pgdrive_amd import engine
    build.build()
    L = engine.load_library()
    syms = _declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(L, s), "libpgdrive_hip.so does not export %s" % s
    assert set(syms) == set(engine.EXPORTS), (set(syms) ^ set(engine.EXPORTS))
    assert L.pgd_version().decode().startswith("pgdrive_hip")


def test_struct_sizes_match_c():
    """sizeof() of every ABI struct, taken from a tiny C program compiled against the header."""
    prog = r'''
#include <stdio.h>
#include "pgdrive_hip.h"
#include "pgd_state_layout.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %d %d %d\n", sizeof(pgd_lane), sizeof(pgd_road), sizeof(pgd_box), sizeof(pgd_map),
         sizeof(pgd_spawn), sizeof(pgd_scenario), sizeof(pgd_config), PGD_NF, PGD_NI, PGD_NEI);
  return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(prog)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = [int(x) for x in out]
    assert sizes[0] == mapdata.LANE_DT.itemsize == 64
    assert sizes[1] == mapdata.ROAD_DT.itemsize == 16
    assert sizes[2] == mapdata.BOX_DT.itemsize == 32
    assert sizes[3] == mapdata.MAP_DT.itemsize
    assert sizes[4] == scenario.SPAWN_DT.itemsize
    assert sizes[5] == scenario.SCEN_DT.itemsize
    assert sizes[6] == C.sizeof(_abi.PgdConfig)
    assert sizes[7:] == [_abi.NF, _abi.NI, _abi.NEI]


def test_obs_dim_matches_reference_layout():
    """274 = 8 ego + 10 navi + 16 neighbours + 240 beams (obs/state_obs.py:17-23,124-130); 18 without lidar."""
    from pgdrive_amd import engine
    L = engine.load_library()
    cfg = _abi.make_config(4)
    assert L.pgd_obs_dim(C.byref(cfg)) == 274 == _abi.obs_dim(cfg)
    cfg = _abi.make_config(4, num_traffic=0, num_lasers=0)
    assert L.pgd_obs_dim(C.byref(cfg)) == 18
    cfg = _abi.make_config(4, num_lasers=72, num_others=0)  # MARL default (multi_agent_pgdrive.py:38)
    assert L.pgd_obs_dim(C.byref(cfg)) == 90


def test_bad_arguments_are_rejected_without_a_gpu():
    from pgdrive_amd import engine
    L = engine.load_library()
    h = C.c_void_p()
    cfg = _abi.make_config(0)
    assert L.pgd_create(C.byref(cfg), 0, None, C.byref(h)) == 1  # PGD_ERR_ARG
    cfg = _abi.make_config(4, num_traffic=100)
    assert L.pgd_create(C.byref(cfg), 0, None, C.byref(h)) == 1
    assert L.pgd_step(None, None, None, None, None, None) == 1
    assert L.pgd_destroy(None) == 1


def test_engine_fails_loudly_without_gpu_or_library(monkeypatch):
    import pytest
    import torch
    from pgdrive_amd import engine
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            engine.Engine(_abi.make_config(1), None, None)
    monkeypatch.setattr(engine, "_LIBH", None)
    with pytest.raises(RuntimeError, match="missing"):
        engine.load_library(path="/nonexistent/libpgdrive_hip.so")


def test_field_offsets_match_c():
    """offsetof() of every field of pgd_config / pgd_spawn / pgd_scenario / pgd_lane / pgd_road / pgd_box / pgd_map in the
    header == the Python mirrors (a size check alone would not notice two swapped fields)."""
    mirrors = {
        "pgd_config": [(n, getattr(_abi.PgdConfig, n).offset) for n, _ in _abi.PgdConfig._fields_],
        "pgd_spawn": [(n, scenario.SPAWN_DT.fields[n][1]) for n in scenario.SPAWN_DT.names],
        "pgd_scenario": [(n, scenario.SCEN_DT.fields[n][1]) for n in scenario.SCEN_DT.names],
        "pgd_lane": [(n, mapdata.LANE_DT.fields[n][1]) for n in mapdata.LANE_DT.names],
        "pgd_road": [(n, mapdata.ROAD_DT.fields[n][1]) for n in mapdata.ROAD_DT.names],
        "pgd_box": [(n, mapdata.BOX_DT.fields[n][1]) for n in mapdata.BOX_DT.names],
        "pgd_map": [(n, mapdata.MAP_DT.fields[n][1]) for n in mapdata.MAP_DT.names],
    }
    alias = {("pgd_road", "frm"): "from"}  # python keyword
    lines = []
    for st, fields in mirrors.items():
        for n, _ in fields:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, n, st, alias.get((st, n), n)))
    prog = "#include <stdio.h>\n#include <stddef.h>\n#include \"pgdrive_hip.h\"\nint main(void) {\n" + "\n".join(lines) + "\n  return 0;\n}\n"
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "o.c")
        open(c, "w").write(prog)
        exe = os.path.join(td, "o")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = dict(line.split() for line in subprocess.check_output([exe]).decode().strip().splitlines())
    n = 0
    for st, fields in mirrors.items():
        for name, off in fields:
            assert int(out["%s.%s" % (st, name)]) == off, (st, name, out["%s.%s" % (st, name)], off)
            n += 1
    assert n > 100


def test_prepared_policy_buffer_size_is_a_pure_function():
    """pgd_mlp_prepared_bytes (no GPU needed): two layers of 32-row chunks x 4 waves x 4 tiles x 2 planes x 64 lanes x 16 bytes, then
    b1, b2, the head's 2 x 256 weights and b3 (4 floats reserved); 0 for an input width the kernels refuse."""
    from pgdrive_amd import engine
    L = engine.load_library()
    chunk = 4 * 4 * 2 * 64 * 16
    assert L.pgd_mlp_prepared_bytes(274) == (9 + 8) * chunk + 4 * (4 * 256 + 4)
    assert L.pgd_mlp_prepared_bytes(256) == (8 + 8) * chunk + 4 * (4 * 256 + 4)
    assert L.pgd_mlp_prepared_bytes(3) == 0 and L.pgd_mlp_prepared_bytes(5000) == 0
