"""Multi-agent roundabout (BASELINE config 5) and intersection on the HIP engine.

Mirrors `MultiAgentRoundaboutEnv` (pgdrive/envs/marl_envs/marl_inout_roundabout.py:133-153) on top of
`MultiAgentPGDrive` (pgdrive/envs/marl_envs/multi_agent_pgdrive.py:58-213): per-agent termination, finished agents stay
`delay_done` steps as static obstacles, new agents ("agent{k}", ever increasing k) are respawned into free 8 m x 3 m
places at the road starts while fewer than `num_agents` are alive and the episode is younger than `horizon`;
`done["__all__"]` ends the episode.

`MultiAgentRoundaboutVecEnv` — N envs, slot-indexed cuda tensors `[N, A, ...]` + flag bits (PGD_F_REPORT / NEW / ALL_DONE).
`MultiAgentRoundaboutEnv`    — one env, the reference's dict-in / dict-out protocol.
`MultiAgentIntersectionVecEnv / MultiAgentIntersectionEnv` — the same machinery on `MAIntersectionMap`
(pgdrive/envs/marl_envs/marl_intersection.py:14-109: first block + 4-way intersection with u-turns, 30 agents).
"""
import numpy as np

from . import _abi, bank, mapdata, scenario
from .spaces import Box, Dict, EnvBase
from .vec_env import merge_config, resolve_map_choice, strip_reference_only_keys

# MULTI_AGENT_PGDRIVE_DEFAULT_CONFIG + MARoundaboutConfig (multi_agent_pgdrive.py:12-55, marl_inout_roundabout.py:15-28)
MA_DEFAULT_CONFIG = dict(
    num_envs=1,
    num_agents=40,
    map_config=dict(exit_length=60, lane_num=2, lane_width=3.5),  # marl_inout_roundabout.py:23
    neighbours_distance=10,  # defined upstream (multi_agent_pgdrive.py:35) and read nowhere: accepted, without effect
    max_agents=None,  # slot capacity per env (default: num_agents); the reference has no cap
    # {"agent{k}": dict(spawn_lane_index=(from, to, lane), spawn_longitude=, spawn_lateral=, destination_node=)}: these agents start
    # where they are told instead of in a drawn spawn slot (multi_agent_pgdrive.py:96-107, spawn_manager.py:58-69)
    target_vehicle_configs=None,
    # the roads agents are (re)spawned on: None = the env's own list (e.g. MARoundaboutConfig.spawn_roads), else `Road`-like
    # objects (start_node / end_node) or (from node name, to node name) pairs
    spawn_roads=None,
    crash_done=True,
    out_of_road_done=True,
    delay_done=25,
    allow_respawn=True,
    horizon=1000,
    vehicle_config=dict(lidar=dict(num_lasers=72, distance=40, num_others=0),
                        side_detector=dict(num_lasers=0, distance=50), lane_line_detector=dict(num_lasers=0, distance=20)),
    cross_yellow_line_done=True,  # only read by the bottleneck env (marl_bottleneck.py:130-136)
    success_reward=10.0,
    out_of_road_penalty=10,
    crash_vehicle_penalty=10,
    crash_object_penalty=10,
    crash_vehicle_cost=1, crash_object_cost=1, out_of_road_cost=0,  # cost_function (multi_agent_pgdrive.py:45-47): out of road is free
    driving_reward=1.0,
    speed_reward=0.1,
    use_lateral=False,
    decision_repeat=5,
    physics_world_step_size=2e-2,
    spawn_variants=16,  # number of pre-drawn initial placements (SpawnManager.reset draws afresh every episode)
    auto_reset=True,
    device=0,
    seed=0,
)


class MultiAgentRoundaboutVecEnv:
    MAP_KIND = "roundabout"
    DEFAULTS = MA_DEFAULT_CONFIG
    PLAIN_REWARD = False
    TOLLGATE = False
    PARKING = False

    @staticmethod
    def _generate_map(mc):
        from . import mapgen
        return mapgen.generate_ma_roundabout(mc["lane_num"], mc["lane_width"], mc["exit_length"])

    def __init__(self, config=None):
        config = {k: v for k, v in (config or {}).items() if k != "is_multi_agent"}  # always True here (multi_agent_pgdrive.py:14)
        self.config = c = merge_config(self.DEFAULTS, strip_reference_only_keys(
            config, keep=("num_agents", "allow_respawn", "delay_done")))
        lid = c["vehicle_config"]["lidar"]
        sd, ld = c["vehicle_config"]["side_detector"], c["vehicle_config"]["lane_line_detector"]
        if not 0 <= lid["num_others"] <= 16:
            raise ValueError("lidar.num_others must be in [0, 16]")
        self.desc = self._generate_map(c["map_config"])
        descs = self.desc if isinstance(self.desc, (list, tuple)) else [self.desc]
        self.map_bank = mapdata.MapBank(list(descs), truncate_succ=True)  # no IDM traffic on the multi-agent maps
        # num_agents = -1: "as many vehicles as possible" (base_env.py:25) -- every spawn slot at the start, and the respawn rule
        # stops counting agents (agent_manager.py:316-323).  The reference then grows without bound; here the slot capacity
        # `max_agents` bounds it (default: the number of spawn slots, at most 64 bodies per env)
        cap = c["max_agents"] or (c["num_agents"] if c["num_agents"] != -1 else None)
        fixed = None
        if c["target_vehicle_configs"]:
            if self.PARKING:
                raise ValueError("target_vehicle_configs: not supported on the parking lot (its spawn manager hands out parking spaces)")
            fixed = {}
            for k, v in c["target_vehicle_configs"].items():
                if not (isinstance(k, str) and k.startswith("agent") and k[5:].isdigit()):
                    raise KeyError("target_vehicle_configs: agent names are 'agent0', 'agent1', ... (got %r)" % (k, ))
                unknown = set(v) - {"spawn_lane_index", "spawn_longitude", "spawn_lateral", "destination_node"}
                if unknown:
                    raise KeyError("target_vehicle_configs[%r]: unknown keys %s" % (k, sorted(unknown)))
                fixed[int(k[5:])] = dict(v)
        if c["spawn_roads"] is not None and self.PARKING:
            raise ValueError("spawn_roads: the parking lot takes in_spawn_roads / out_spawn_roads from its map (marl_parking_lot.py:15-24)")
        self.scen_bank = scenario.MarlScenarioBank(self.desc, c["num_agents"], capacity=cap, n_variants=c["spawn_variants"],
                                                   seed=c["seed"], kind=self.MAP_KIND, fixed=fixed, spawn_roads=c["spawn_roads"])
        cap = self.scen_bank.A
        self.num_envs, self.A = int(c["num_envs"]), cap
        self.cfg = _abi.make_config(
            self.num_envs, num_agents=cap, num_traffic=self.scen_bank.B, num_lasers=lid["num_lasers"],
            # neighbour rows: the neighbour's own state vector (LidarStateObservationMARound, marl_inout_roundabout.py:82-105);
            # the tollgate env observes through TollGateObservation, a LidarStateObservation: 4 relative floats per neighbour
            num_others=lid["num_others"], others_state=lid["num_others"] > 0 and not self.TOLLGATE,
            lidar_dist=lid["distance"], dt=c["physics_world_step_size"], decision_repeat=c["decision_repeat"],
            auto_reset=c["auto_reset"], resample_scenario=1, horizon=c["horizon"] or 0, seed=c["seed"],
            success_reward=c["success_reward"], out_of_road_penalty=c["out_of_road_penalty"],
            crash_vehicle_penalty=c["crash_vehicle_penalty"], crash_object_penalty=c["crash_object_penalty"],
            driving_reward=c["driving_reward"], speed_reward=c["speed_reward"], use_lateral=c["use_lateral"],
            multi_agent=True, crash_done=c["crash_done"], out_of_road_done=c["out_of_road_done"],
            allow_respawn=c["allow_respawn"], delay_done=c["delay_done"], agent_limit=cap if self.scen_bank.infinite else c["num_agents"],
            respawn_places=self.scen_bank.P, respawn_dests=self.scen_bank.Dn,
            side_lasers=sd["num_lasers"] if sd["distance"] > 0 else 0, side_dist=sd["distance"],
            lane_line_lasers=ld["num_lasers"] if ld["distance"] > 0 else 0, lane_line_dist=ld["distance"],
            plain_reward=self.PLAIN_REWARD, cross_yellow_line_done=c["cross_yellow_line_done"], tollgate=self.TOLLGATE,
            overspeed_penalty=c.get("overspeed_penalty", 0.5), min_pass_steps=c["vehicle_config"].get("min_pass_steps", 30),
            parking=self.PARKING, enable_reverse=c["vehicle_config"].get("enable_reverse", False)
        )
        from .engine import Engine
        self.engine = Engine(self.cfg, self.map_bank, self.scen_bank, device=c["device"])
        self.obs_dim = self.engine.D
        self.single_observation_space = Box(-0.0, 1.0, (self.obs_dim, ), np.float32)
        self.single_action_space = Box(-1.0, 1.0, (2, ), np.float32)
        self._rng = np.random.RandomState(c["seed"])

    def reset(self):
        """SpawnManager.reset draws the placement afresh every episode (spawn_manager.py:68-101); here an env is dealt one of the
        `spawn_variants` pre-drawn placements -- never the one it was dealt at its last reset() (the reference's own
        test_randomize_spawn_place expects every agent somewhere else after a reset).  Returns obs [N, A, D]; rows of empty seats are
        zero, and a row that is not due stays as it was left: callers that post-process rows in place should copy first (see
        Engine.step)."""
        n = len(self.scen_bank.scenarios)
        ids = self._rng.randint(0, n, size=self.num_envs).astype(np.int32)
        if n > 1 and getattr(self, "_last_ids", None) is not None:
            same = ids == self._last_ids
            ids[same] = (ids[same] + 1 + self._rng.randint(0, n - 1, size=int(same.sum()))) % n
        self._last_ids = ids.copy()
        return self.engine.reset(ids)

    def step(self, actions):
        """actions [N, A, 2] cuda float32 (rows of slots without an active agent are ignored).  Returns the engine's own buffers
        (obs [N, A, D], reward, done, flags), not copies: the row of a seat that is not due reads zero and is written ONCE per
        buffer (Engine.step) -- clone before editing rows in place, or create the env with PGD_NO_ROWZ=1 in the environment."""
        return self.engine.step(actions.contiguous())

    # -- asynchronous env groups (pgd_set_groups / pgd_step_group): a multi-agent step is two launches, a step kernel that waits for
    # memory half of its life and an observation kernel; as two halves on two streams the step of one half runs beside the observation of
    # the other (40 seats, 4096 envs: 81 -> 93 M env-steps/s, bench.py row c5_40x72_two_groups; examples/marl_env_groups.py) ----------
    def set_groups(self, n_groups):
        """Split the envs into `n_groups` equal contiguous groups with their own streams (`engine.group_streams`)."""
        self.engine.set_groups(n_groups)

    def step_group(self, g, actions):
        """Step the envs of group g only, asynchronously on the group's stream; `actions` is the full [N, A, 2] tensor (the group reads
        its own rows).  Returns views of the group's rows of the engine's buffers: (obs, reward, done, flags) -- work that reads them
        belongs on `engine.group_streams[g]`, or behind `group_sync(g)`."""
        return self.engine.step_group(g, actions.contiguous())

    def group_sync(self, g):
        self.engine.group_sync(g)

    def group_slice(self, g):
        return self.engine.group_slice(g)

    def slot_table(self):
        """Host copy of (status, agent id) per slot: ([N, A] int, [N, A] int)."""
        f, i, _ = self.engine.get_state()
        return i[_abi.SI["STATUS"]], f[_abi.SF["AGENT_ID"]].astype(np.int64)

    def close(self):
        if getattr(self, "engine", None) is not None:
            self.engine.close()
            self.engine = None


class MultiAgentIntersectionVecEnv(MultiAgentRoundaboutVecEnv):
    """MultiAgentIntersectionEnv (marl_intersection.py:65-109) batched: MAIntersectionConfig = 30 agents, 2 lanes, exits 60 m."""
    MAP_KIND = "intersection"
    DEFAULTS = dict(MA_DEFAULT_CONFIG, num_agents=30)

    @staticmethod
    def _generate_map(mc):
        from . import mapgen
        return mapgen.generate_ma_intersection(mc["lane_num"], mc["lane_width"], mc["exit_length"])


class MultiAgentBottleneckVecEnv(MultiAgentRoundaboutVecEnv):
    """MultiAgentBottleneckEnv (marl_bottleneck.py:11-137) batched: 4-lane road merging into a 1-lane neck and splitting
    again, 20 agents, side (4 x 50 m) and lane-line (4 x 20 m) detector fans in the observation, its own reward / out-of-road
    variants, destinations from Navigation's default rule."""
    MAP_KIND = "bottleneck"
    PLAIN_REWARD = True
    DEFAULTS = dict(
        MA_DEFAULT_CONFIG, num_agents=20,
        map_config=dict(exit_length=60, lane_width=3.5, bottle_lane_num=4, neck_lane_num=1, neck_length=20),
        vehicle_config=dict(lidar=dict(num_lasers=72, distance=40, num_others=0), side_detector=dict(num_lasers=4, distance=50),
                            lane_line_detector=dict(num_lasers=4, distance=20)),
    )

    @staticmethod
    def _generate_map(mc):
        from . import mapgen
        return mapgen.generate_ma_bottleneck(mc["lane_width"], mc["exit_length"], mc["bottle_lane_num"], mc["neck_lane_num"],
                                             mc["neck_length"])


class MultiAgentTollgateVecEnv(MultiAgentRoundaboutVecEnv):
    """MultiAgentTollgateEnv (marl_tollgate.py:14-279) batched: a 3-lane road fanning out to an 8-lane toll plaza (booths on
    every odd lane: crash_building) and back, 40 agents.  Observation = vehicle state with 72 side + 4 lane-line beams, no
    navigation block, 72 lidar beams x 20 m and [inside the plaza, stayed longer than min_pass_steps]; inside the plaza
    driving faster than the lane limit costs `overspeed_penalty`, and an agent that crossed it in fewer than
    `min_pass_steps` steps is terminated."""
    MAP_KIND = "tollgate"
    PLAIN_REWARD = True
    TOLLGATE = True
    DEFAULTS = dict(
        MA_DEFAULT_CONFIG, num_agents=40, speed_reward=0.0, overspeed_penalty=0.5,
        map_config=dict(exit_length=70, lane_width=3.5, lane_num=3, toll_lane_num=8, toll_length=10),
        vehicle_config=dict(lidar=dict(num_lasers=72, distance=20, num_others=0), side_detector=dict(num_lasers=72, distance=20),
                            lane_line_detector=dict(num_lasers=4, distance=20), min_pass_steps=30),
    )

    @staticmethod
    def _generate_map(mc):
        from . import mapgen
        return mapgen.generate_ma_tollgate(mc["lane_num"], mc["lane_width"], mc["exit_length"], mc["toll_lane_num"],
                                           mc["toll_length"])


class MultiAgentParkingLotVecEnv(MultiAgentRoundaboutVecEnv):
    """MultiAgentParkingLotEnv (marl_parking_lot.py:15-222) batched: a one-lane road through a lot with 8 perpendicular
    parking spaces and a T-intersection behind it, 10 agents that can reverse; agents entering from a road drive to a free
    parking space, agents starting in a space leave through a random access road."""
    MAP_KIND = "parking"
    PARKING = True
    DEFAULTS = dict(
        MA_DEFAULT_CONFIG, num_agents=10, parking_space_num=8, map_config=dict(exit_length=20, lane_width=3.5, lane_num=1),
        vehicle_config=dict(lidar=dict(num_lasers=72, distance=40, num_others=0), side_detector=dict(num_lasers=0, distance=50),
                            lane_line_detector=dict(num_lasers=0, distance=20), enable_reverse=True),
    )

    @staticmethod
    def _generate_map(mc):
        from . import mapgen
        assert mc["lane_num"] == 1, "the parking lot needs a one-lane road (parking_lot.py:29)"
        return mapgen.generate_ma_parking_lot(mc["lane_width"], mc["exit_length"], mc.get("parking_space_num", 8))

    def __init__(self, config=None):
        cfg = dict(config or {})
        n = cfg.get("parking_space_num", 8)
        assert n % 2 == 0 and n >= 4, "number of parking spaces must be a multiple of 2, at least 4"  # marl_parking_lot.py:146-147
        mc = dict(cfg.get("map_config", {}))
        mc["parking_space_num"] = n
        cfg["map_config"] = mc
        self.DEFAULTS = dict(self.DEFAULTS, map_config=dict(self.DEFAULTS["map_config"], parking_space_num=8))
        super().__init__(cfg)


class VehicleHandle:
    """What the reference's own multi-agent tests reach for on `env.vehicles[name]` (a BaseVehicle): pose and speed, the contact
    flags of the last step, `set_position` (a teleport: tests put agents on top of each other or on their destination),
    `set_static`, and the end of the final lane of the agent's route (`navigation.final_lane.end`).  A host-side convenience over
    pgd_get_state / pgd_set_state (one round trip per call: single-env test surface, not the batched hot path)."""

    def __init__(self, env, name, slot):
        self._env, self.name, self.slot = env, name, slot

    def _state(self):
        return self._env.vec.engine.get_state()

    @property
    def position(self):
        f, _, _ = self._state()
        return np.array([f[_abi.SF["X"], 0, self.slot], f[_abi.SF["Y"], 0, self.slot]], dtype=np.float64)

    @property
    def heading_theta(self):
        return float(self._state()[0][_abi.SF["THETA"], 0, self.slot])

    @property
    def speed(self):
        """km/h, as BaseVehicle.speed (base_vehicle.py:394-401)."""
        return abs(float(self._state()[0][_abi.SF["SPEED"], 0, self.slot])) * 3.6

    def _spawn_record(self):
        _, i, ei = self._state()
        bank_ = self._env.vec.scen_bank
        return bank_.spawns[int(ei[_abi.EI["SCEN"], 0]) * bank_.stride + int(i[_abi.SI["SPAWN"], 0, self.slot])]

    @property
    def length(self):
        return float(self._spawn_record()["length"])

    @property
    def width(self):
        return float(self._spawn_record()["width"])

    @property
    def final_lane(self):
        """The lane description the agent's route ends on (Navigation.final_lane, navigation.py:123-148)."""
        rec = self._spawn_record()
        descs = self._env.vec.map_bank.descs
        _, _, ei = self._state()
        m = int(self._env.vec.scen_bank.scenarios[int(ei[_abi.EI["SCEN"], 0])]["map"])
        return descs[m]["lanes"][int(rec["dest_lane"])]

    def _flags(self):
        return int(self._state()[1][_abi.SI["VFLAGS"], 0, self.slot])

    @property
    def crash_vehicle(self):
        return bool(self._flags() & _abi.F_CRASH_VEHICLE)

    @property
    def out_of_road(self):
        return bool(self._flags() & _abi.F_OUT_OF_ROAD)

    def set_position(self, pos, height=None):
        """Teleport (BaseVehicle.set_position, base_vehicle.py:418-425): heading and speed stay; the next step localises the
        vehicle where it now stands."""
        f, i, ei = self._state()
        f[_abi.SF["X"], 0, self.slot], f[_abi.SF["Y"], 0, self.slot] = float(pos[0]), float(pos[1])
        self._env.vec.engine.set_state(f, i, ei)

    def set_static(self, flag=True):
        """BaseVehicle.set_static: the body no longer moves whatever action it is given.  Here: the env overrides the slot's action
        with [0, 0] and the speed is zeroed once (a standing car with [0, 0] stays where it is: rolling brake, no reverse)."""
        if flag:
            self._env._static.add(self.slot)
            f, i, ei = self._state()
            if f[_abi.SF["SPEED"], 0, self.slot] != 0.0:
                f[_abi.SF["SPEED"], 0, self.slot] = 0.0
                self._env.vec.engine.set_state(f, i, ei)
        else:
            self._env._static.discard(self.slot)


class MultiAgentRoundaboutEnv(EnvBase):  # (gym.Env when gym is importable, like the reference's MultiAgentPGDrive)
    """Dict protocol of the reference: keys "agent{k}"; done has "__all__" (multi_agent_pgdrive.py:126-150)."""
    VEC = MultiAgentRoundaboutVecEnv

    def __init__(self, config=None):
        cfg = dict(config or {})
        cfg["num_envs"] = 1
        cfg.setdefault("auto_reset", False)
        # Seats.  The engine never re-uses a seat in the step in which its agent reported (the terminal row must survive), so with
        # exactly num_agents seats a step in which EVERY agent finishes leaves nobody to respawn into and reads as `__all__` -- the
        # reference respawns in that very step (multi_agent_pgdrive.py:126-150; its own tests finish both of two agents at once and
        # expect the episode to go on).  The dict env therefore keeps spare seats unless told otherwise: as many as agents, bodies per
        # env capped at 52 (the sub-step contact test of larger waves is coarser).  `num_agents` still bounds how
        # many are alive at a time (agent_limit).
        n = cfg.get("num_agents", self.VEC.DEFAULTS["num_agents"])
        if cfg.get("max_agents") is None and n is not None and n > 0:
            cfg["max_agents"] = min(2 * n, max(n, 44))
        self.vec = self.VEC(cfg)
        self.config = self.vec.config
        import torch
        self._torch = torch
        self._slots = {}  # agent name -> slot
        self._static = set()  # slots frozen by VehicleHandle.set_static
        self.episode_steps = 0

    @property
    def num_agents(self):
        """How many agents may be alive at a time (BaseEnv.num_agents)."""
        return int(self.vec.scen_bank.num_agents)

    @property
    def vehicles(self):
        """name -> VehicleHandle of the agents that act in the next step (BaseEnv.vehicles)."""
        return {k: VehicleHandle(self, k, s) for k, s in self._slots.items()}

    def finish(self, agent_name):
        """AgentManager.finish (agent_manager.py:134-153): the agent stops acting at once; it stays `delay_done` steps as a static
        body (none with delay_done = 0) and its seat may be respawned into."""
        s = self._slots.pop(agent_name)
        f, i, ei = self.vec.engine.get_state()
        dd = int(self.vec.cfg.delay_done)
        i[_abi.SI["STATUS"], 0, s] = _abi.ST_DYING if dd > 0 else _abi.ST_EMPTY
        i[_abi.SI["TIMER"], 0, s] = dd
        self.vec.engine.set_state(f, i, ei)
        self._static.discard(s)

    @property
    def observation_space(self):
        return Dict({k: self.vec.single_observation_space for k in self._slots})

    @property
    def action_space(self):
        return Dict({k: self.vec.single_action_space for k in self._slots})

    def _refresh_slots(self):
        status, ids = self.vec.slot_table()
        self._slots = {"agent%d" % ids[0, s]: s for s in range(self.vec.A) if status[0, s] == _abi.ST_ACTIVE}

    def reset(self):
        obs = self.vec.reset()[0].cpu().numpy()
        self._refresh_slots()
        self._static.clear()
        self.episode_steps = 0
        return {k: obs[s] for k, s in self._slots.items()}

    def step(self, actions):
        a = np.zeros((1, self.vec.A, 2), dtype=np.float32)
        for k, s in self._slots.items():  # extra keys are ignored like base_env.py:205-212
            if k in actions and s not in self._static:
                a[0, s] = np.asarray(actions[k], dtype=np.float32)
        f0, i0, _ = self.vec.engine.get_state()  # episode_reward / episode_length of the agents that are about to act
        obs, rew, done, flags = self.vec.step(self._torch.from_numpy(a).to(self.vec.engine.device))
        self.vec.engine.sync()
        obs, rew, done = obs[0].cpu().numpy(), rew[0].cpu().numpy(), done[0].cpu().numpy()
        f1, i1, _ = self.vec.engine.get_state()
        fl = flags[0].cpu().numpy().astype(np.uint32)
        self.episode_steps += 1
        o, r, d, info = {}, {}, {}, {}
        for k, s in self._slots.items():  # agents that acted this step
            assert fl[s] & _abi.F_REPORT
            o[k], r[k], d[k] = obs[s], float(rew[s]), bool(done[s])
            info[k] = dict(
                arrive_dest=bool(fl[s] & _abi.F_ARRIVE), out_of_road=bool(fl[s] & _abi.F_OUT_OF_ROAD),
                crash_vehicle=bool(fl[s] & _abi.F_CRASH_VEHICLE), crash=bool(fl[s] & _abi.F_CRASH_VEHICLE),
                crash_object=bool(fl[s] & _abi.F_CRASH_OBJECT), crash_building=bool(fl[s] & _abi.F_CRASH_BUILDING),
                max_step=bool(fl[s] & _abi.F_MAX_STEP), step_reward=float(rew[s]),
                # cost_function (pgdrive_env.py:197-207): out of road, else crash_vehicle, else crash_object
                cost=(self.config["out_of_road_cost"] if fl[s] & _abi.F_OUT_OF_ROAD else
                      self.config["crash_vehicle_cost"] if fl[s] & _abi.F_CRASH_VEHICLE else
                      self.config["crash_object_cost"] if fl[s] & _abi.F_CRASH_OBJECT else 0),
                # _get_step_return (base_env.py:335-339) and BaseVehicle.after_step (base_vehicle.py:255-273)
                episode_reward=float(f0[_abi.SF["EP_REWARD"], 0, s]) + float(rew[s]),
                episode_length=int(i0[_abi.SI["RLANE"], 0, s]) + 1,
                raw_action=(float(a[0, s, 0]), float(a[0, s, 1])),
            )
            if i1[_abi.SI["SPAWN"], 0, s] == i0[_abi.SI["SPAWN"], 0, s] and f1[_abi.SF["AGENT_ID"], 0, s] == f0[_abi.SF["AGENT_ID"], 0, s]:
                info[k].update(velocity=abs(float(f1[_abi.SF["SPEED"], 0, s])) * 3.6, steering=float(f1[_abi.SF["STEER"], 0, s]),
                               acceleration=float(f1[_abi.SF["THROTTLE"], 0, s]))  # the slot still holds this agent
        all_done = bool(fl[0] & _abi.F_ALL_DONE)
        self._refresh_slots()
        self._static &= set(self._slots.values())  # a seat that changed hands is no longer frozen
        if not all_done:
            status, ids = self.vec.slot_table()
            for s in range(self.vec.A):  # respawned agents: obs of the newcomer, reward 0, not done
                if fl[s] & _abi.F_NEW and not fl[s] & _abi.F_REPORT:
                    k = "agent%d" % ids[0, s]
                    o[k], r[k], d[k], info[k] = obs[s], 0.0, False, {}
        d["__all__"] = all_done
        if all_done:
            for k in list(d.keys()):
                d[k] = True
        return o, r, d, info

    def close(self):
        self.vec.close()


class MultiAgentIntersectionEnv(MultiAgentRoundaboutEnv):
    """Dict protocol on the intersection map (marl_intersection.py:65-109)."""
    VEC = MultiAgentIntersectionVecEnv


class MultiAgentBottleneckEnv(MultiAgentRoundaboutEnv):
    """Dict protocol on the bottleneck map (marl_bottleneck.py:70-137)."""
    VEC = MultiAgentBottleneckVecEnv


class MultiAgentTollgateEnv(MultiAgentRoundaboutEnv):
    """Dict protocol on the toll plaza map (marl_tollgate.py:163-279)."""
    VEC = MultiAgentTollgateVecEnv


class MultiAgentParkingLotEnv(MultiAgentRoundaboutEnv):
    """Dict protocol on the parking-lot map (marl_parking_lot.py:133-222)."""
    VEC = MultiAgentParkingLotVecEnv


class MultiAgentPGDriveVecEnv(MultiAgentRoundaboutVecEnv):
    """MultiAgentPGDrive itself (multi_agent_pgdrive.py:12-213) batched: the generic multi-agent env over generated PG maps
    (`start_seed` .. `start_seed + environment_num`), 15 agents spawned on the straight of the first block ('>>' -> '>>>':
    5 slots x 3 lanes), destinations from Navigation's default rule, respawn into the same slots."""
    MAP_KIND = "pg"
    DEFAULTS = dict(MA_DEFAULT_CONFIG, num_agents=15, start_seed=0, environment_num=1, map=3,
                    map_config=dict(exit_length=50, lane_num=3, lane_width=3.5, type="block_num", config=None))

    def _generate_map(self, mc):
        c = self.config
        m = resolve_map_choice(c)
        kw = dict(block_num=m) if isinstance(m, int) else dict(block_seq=m, block_num=None)
        seeds = range(c["start_seed"], c["start_seed"] + c["environment_num"])
        return bank.get_descriptions(seeds, mc["lane_num"], mc["lane_width"], mc["exit_length"], **kw)


class MultiAgentPGDrive(MultiAgentRoundaboutEnv):
    """Dict protocol of the generic multi-agent env (multi_agent_pgdrive.py:58-213)."""
    VEC = MultiAgentPGDriveVecEnv
