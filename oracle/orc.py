"""ctypes wrapper of the CPU oracle liborc.so (TEST INFRASTRUCTURE: tests/, smoke(), bench cpu_baseline only)."""
import ctypes as C
import os
import subprocess

import numpy as np

from pgdrive_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "pgd_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liborc.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(_abi.PgdConfig)]
        for name in ("orc_wrap_to_pi", "orc_not_zero", "orc_norm", "orc_clip", "orc_lane_heading", "orc_lane_distance",
                     "orc_pid", "orc_ray_box", "orc_idm_acc", "orc_heading_diff"):
            getattr(L, name).restype = C.c_double
        L.orc_wrap_to_pi.argtypes = [C.c_double]
        L.orc_not_zero.argtypes = [C.c_double, C.c_double]
        L.orc_norm.argtypes = [C.c_double, C.c_double]
        L.orc_clip.argtypes = [C.c_double] * 3
        L.orc_lane_heading.argtypes = [C.c_void_p, C.c_double]
        L.orc_lane_distance.argtypes = [C.c_void_p, C.c_double, C.c_double]
        for name in ("orc_lane_heading_f64", "orc_lane_distance_f64"):
            getattr(L, name).restype = C.c_double
        L.orc_lane_heading_f64.argtypes = [C.c_void_p, C.c_double]
        L.orc_lane_distance_f64.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.orc_lane_local_f64.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.orc_lane_position_f64.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.orc_upload_tables_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_upload_spawns_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_upload_config_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_lane_local.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.orc_lane_position.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.orc_pid.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_projection.argtypes = [C.c_double] * 4 + [C.c_void_p]
        L.orc_ray_box.argtypes = [C.c_double] * 9
        L.orc_obb_overlap.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_bicycle.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_idm_acc.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_topdown_snap.restype = C.c_double
        L.orc_topdown_snap.argtypes = [C.c_double]
        L.orc_energy_step.restype = C.c_double
        L.orc_energy_step.argtypes = [C.c_double] * 3
        L.orc_action_forces.argtypes = [C.c_double] * 5 + [C.c_int, C.c_void_p]
        L.orc_before_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        L.orc_navi_info.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_heading_diff.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_localize.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_state_check.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_state_check.restype = C.c_uint
        L.orc_idm_act.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_find_front_back.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_dist_left_right.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_reward_done.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_update_checkpoints.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]
        L.orc_step_range.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.orc_step_mt.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.orc_run_mt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.orc_rng.restype = C.c_uint32
        L.orc_rng.argtypes = [C.c_uint32] * 4
        for name in ("orc_upload_maps", "orc_upload_scenarios", "orc_destroy", "orc_get_state", "orc_set_state",
                     "orc_reset", "orc_observe", "orc_refresh", "orc_step"):
            getattr(L, name).argtypes = None
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    """Same call surface as pgdrive_amd.engine.Engine, float64 on the CPU."""
    def __init__(self, cfg, bank, scen, f64=False):
        """f64: the tables keep the float64 values of the map description (bank.lanes64 / boxes64 / maps64) and, when the scenario
        bank carries them (scen.spawns64 [n, 12]), of the spawn records, instead of the ABI's float32 records -- the path the
        reference-generated goldens are held to at 1e-9 (tests/test_oracle_golden.py).  Default: the float32 records, i.e. the
        very numbers the engine is given (GPU parity tests)."""
        self.L = lib()
        self.cfg = cfg
        self.N, self.A, self.V = cfg.num_envs, cfg.num_agents, cfg.num_agents + cfg.num_traffic
        self.D = _abi.obs_dim(cfg)
        self.h = C.c_void_p(self.L.orc_create(C.byref(cfg)))
        self.bank, self.scen = bank, scen
        assert scen.V == self.V
        self.L.orc_upload_maps(
            self.h, _p(bank.maps), len(bank.maps), _p(bank.lanes), len(bank.lanes), _p(bank.roads), len(bank.roads),
            _p(bank.boxes), len(bank.boxes), _p(bank.cell_start), len(bank.cell_start), _p(bank.cell_items),
            len(bank.cell_items)
        )
        self.L.orc_upload_scenarios(self.h, _p(scen.scenarios), len(scen.scenarios), _p(scen.spawns))
        if f64:
            # the float fields of the config as the reference holds them: a float32 field that is the rounding of a short decimal
            # (0.1f, 0.02f) is taken as that decimal
            names = ("lidar_dist", "dt", "success_reward", "out_of_road_penalty", "crash_vehicle_penalty", "crash_object_penalty",
                     "driving_reward", "speed_reward", "side_dist", "lane_line_dist", "overspeed_penalty")
            cd = np.array([float(np.format_float_positional(np.float32(getattr(cfg, k)), unique=True, trim="-")) for k in names],
                          dtype=np.float64)
            assert self.L.orc_upload_config_f64(self.h, _p(cd), len(cd)) == 0
            assert self.L.orc_upload_tables_f64(self.h, _p(bank.lanes64), len(bank.lanes64), _p(bank.boxes64), len(bank.boxes64),
                                                _p(bank.maps64), len(bank.maps64)) == 0
            sp64 = getattr(scen, "spawns64", None)
            if sp64 is not None:
                sp64 = np.ascontiguousarray(sp64, dtype=np.float64)
                assert sp64.shape == (len(scen.spawns), 12) and self.L.orc_upload_spawns_f64(self.h, _p(sp64), len(sp64)) == 0

    def reset(self, scen_ids, env_ids=None):
        scen_ids = np.ascontiguousarray(scen_ids, dtype=np.int32)
        env_ids = None if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int32)
        obs = np.zeros((self.N, self.A, self.D), dtype=np.float64)
        self.L.orc_reset(self.h, _p(env_ids), _p(scen_ids), C.c_int(len(scen_ids)), _p(obs))
        return obs

    def observe(self):
        obs = np.zeros((self.N, self.A, self.D), dtype=np.float64)
        self.L.orc_observe(self.h, _p(obs))
        return obs

    def enable_topdown(self, td_cfg):
        import ctypes as C
        self.td_cfg = td_cfg
        self.L.orc_topdown_enable.argtypes = [C.c_void_p, C.c_void_p]
        self.L.orc_observe_topdown.argtypes = [C.c_void_p, C.c_void_p]
        assert self.L.orc_topdown_enable(self.h, C.byref(td_cfg)) == 0

    def observe_topdown(self):
        R, Cn = self.td_cfg.resolution, (3 if self.td_cfg.mode == 1 else 2 + self.td_cfg.frame_stack)
        img = np.zeros((self.N, R, R, Cn), dtype=np.float64)
        assert self.L.orc_observe_topdown(self.h, _p(img)) == 0
        return img

    def refresh(self):
        self.L.orc_refresh(self.h)

    def step(self, actions, env_range=None, threads=1):
        actions = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.N, self.A, 2)
        obs = np.zeros((self.N, self.A, self.D), dtype=np.float64)
        rew = np.zeros((self.N, self.A), dtype=np.float64)
        done = np.zeros((self.N, self.A), dtype=np.uint8)
        flags = np.zeros((self.N, self.A), dtype=np.uint32)
        if threads > 1:
            self.L.orc_step_mt(self.h, int(threads), _p(actions), _p(obs), _p(rew), _p(done), _p(flags))
        elif env_range is None:
            self.L.orc_step(self.h, _p(actions), _p(obs), _p(rew), _p(done), _p(flags))
        else:
            self.L.orc_step_range(self.h, env_range[0], env_range[1], _p(actions), _p(obs), _p(rew), _p(done), _p(flags))
        return obs, rew, done, flags

    def run(self, action_ring, steps, threads):
        """`steps` steps inside one OpenMP region (bench.py all-cores baseline); returns the last step's outputs."""
        ring = np.ascontiguousarray(action_ring, dtype=np.float32).reshape(-1, self.N, self.A, 2)
        obs = np.zeros((self.N, self.A, self.D), dtype=np.float64)
        rew = np.zeros((self.N, self.A), dtype=np.float64)
        done = np.zeros((self.N, self.A), dtype=np.uint8)
        flags = np.zeros((self.N, self.A), dtype=np.uint32)
        self.L.orc_run_mt(self.h, int(threads), int(steps), _p(ring), ring.shape[0], _p(obs), _p(rew), _p(done), _p(flags))
        return obs, rew, done, flags

    def get_state(self):
        f = np.zeros((_abi.NF, self.N, self.V), dtype=np.float64)
        i = np.zeros((_abi.NI, self.N, self.V), dtype=np.int32)
        ei = np.zeros((_abi.NEI, self.N), dtype=np.int32)
        self.L.orc_get_state(self.h, _p(f), _p(i), _p(ei))
        return f, i, ei

    def set_state(self, f, i, ei):
        f = np.ascontiguousarray(f, dtype=np.float64)
        i = np.ascontiguousarray(i, dtype=np.int32)
        ei = np.ascontiguousarray(ei, dtype=np.int32)
        self.L.orc_set_state(self.h, _p(f), _p(i), _p(ei))

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
