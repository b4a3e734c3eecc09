"""Batched gym-style surface over the HIP step engine.

`PGDriveVecEnv` mirrors the reference's env surface (pgdrive/envs/base_env.py:184-193 step, :269-290 reset,
:403-425 spaces; defaults pgdrive/envs/pgdrive_env.py:22-109) for N environments at once: observations keep the
reference layout (obs/state_obs.py), config keys keep the reference names, unknown keys raise
KeyError like Config.update(allow_add_new_key=False) (utils/config.py:115-125).
"""
import copy

import numpy as np

from . import _abi, bank, mapdata, scenario
from .spaces import Box, MultiDiscrete

# Reference defaults this build honours (pgdrive_env.py:22-109, base_env.py:19-90)
DEFAULT_CONFIG = dict(
    num_envs=1,
    start_seed=0,  # the reference's defaults (base_env.py:20-21, pgdrive_env.py:24-25); gym ids set their own ranges
    environment_num=1,  # (pgdrive_amd.env.ENV_IDS = register.py:5-38, e.g. PGDrive-v0: start_seed 1000, 100 maps)
    map=3,
    # `type` / `config`: the long form of `map` (pgdrive_env.py:31-37, base_map.py:16-35): "block_num" + an int or "block_sequence"
    # + a string of block ids; `config` = None leaves the choice to `map`
    map_config=dict(lane_width=3.5, lane_num=3, exit_length=50, type="block_num", config=None),
    random_lane_width=False,  # a lane width per map seed, uniform in [3.0, 4.5) (map_manager.py:157-163)
    random_lane_num=False,  # a lane count per map seed, randint(2, 3) as upstream (map_manager.py:164-167)
    traffic_density=0.1,
    traffic_mode="trigger",  # "trigger" | "hybrid" (same as trigger upstream) | "respawn" (traffic_manager.py:19-27)
    random_traffic=False,  # True: traffic layout drawn from the env seed instead of the map seed (traffic_manager.py:348-350)
    auto_termination=False,  # done after 250 * num_blocks steps (base_env.py:318)
    discrete_action=False,
    discrete_steering_dim=5,
    discrete_throttle_dim=5,
    max_traffic_vehicles=16,  # slot cap per env (the reference has no cap)
    random_agent_model=False,  # a random vehicle type per episode + its LENGTH / WIDTH in the observation (base_env.py:29)
    accident_prob=0.0,  # TrafficObjectManager (object_manager.py:40-124) when the env registers it (SafePGDriveEnv)
    max_traffic_objects=40,  # extra slots for cones / tripods / barriers / broken-down vehicles when accident_prob > 0
    safe_rl_env=False,  # SafePGDriveEnv.done_function: crashes are not terminal (safe_pgdrive_env.py:49-56)
    crash_vehicle_cost=1.0, crash_object_cost=1.0, out_of_road_cost=1.0,  # cost_function (pgdrive_env.py:197-207)
    decision_repeat=5,
    physics_world_step_size=2e-2,
    horizon=None,
    vehicle_config=dict(
        lidar=dict(num_lasers=240, distance=50, num_others=4, gaussian_noise=0.0, dropout_prob=0.0),
        side_detector=dict(num_lasers=0, distance=50, gaussian_noise=0.0, dropout_prob=0.0),
        lane_line_detector=dict(num_lasers=0, distance=20, gaussian_noise=0.0, dropout_prob=0.0),
        action_check=False,  # step() asserts that the actions lie in the action space (base_vehicle.py:231-236)
        enable_reverse=False,  # negative throttle drives backwards instead of braking (base_vehicle.py:370-373)
        spawn_lane_index=None,  # (from node, to node, lane) by node names; None = ('>', '>>', 0) (pgdrive_env.py:77)
        spawn_longitude=5.0,
        spawn_lateral=0.0,
        destination_node=None,  # node name; None = a seeded random socket of the last block (navigation.py:99-121)
        vehicle_model="default",
        increment_steering=False,
    ),
    success_reward=10.0,
    out_of_road_penalty=5.0,
    crash_vehicle_penalty=5.0,
    crash_object_penalty=5.0,
    driving_reward=1.0,
    speed_reward=0.1,
    use_lateral=False,
    out_of_route_done=False,
    auto_reset=True,
    resample_scenario=True,  # a new seed per episode like _reset_global_seed (base_env.py:451-458)
    device=0,
    seed=0,
    # top-down multi-channel image observation instead of the state + lidar vector (TopDownPGDriveEnv, envs/top_down_env.py:8-42,
    # obs/top_down_obs_multi_channel.py); pgdrive_amd/csrc/pgd_topdown.h states what exactly is drawn
    use_topdown=False, frame_stack=3, post_stack=5, frame_skip=5, resolution_size=84, distance=30, rgb_clip=True,
    topdown_single_frame=False,  # TopDownObservation instead of TopDownMultiChannel: one RGB frame (TopDownSingleFramePGDriveEnv)
    idm_steer_lag=0.0,  # NOT a reference key (an opt-in of this build, default off): time constant [s] of a first-order lag on the steering
                        # IDM-driven vehicles apply -- stands in for the yaw dynamics the kinematic bicycle lacks (include/pgdrive_hip.h
                        # pgd_config::idm_steer_lag; 0.2 settles the traffic on its lane axis).  Such an engine runs the general step kernel
    jit_step_kernel=False,  # NOT a reference key: True builds, in a background thread, a step kernel with this env's configuration
                            # compiled in (pgdrive_amd/jit.py; ~8 s of hipcc once per configuration, cached): configurations
                            # without an instantiation in the library then step 12 - 17 % faster
    IDM_agent=False,  # the ego is driven by IDMPolicy along its route, step()'s actions are ignored (base_env.py:30, agent_manager.py:79)
    map_bank=None,  # path of a pre-generated description bank; None -> generate with our BIG (pgdrive_amd/mapgen.py)
)


# Reference config keys this engine does not act on (BASE_DEFAULT_CONFIG base_env.py:19-90, PGDriveEnv_DEFAULT_CONFIG
# pgdrive_env.py:22-109), so that a config written for the reference can be passed unchanged.
# VISUAL: rendering / window / camera / debugging switches without any effect on what step() returns: dropped.
VISUAL_KEYS = {
    "debug", "fast", "cull_scene", "controller", "use_chase_camera_follow_lane", "camera_height", "camera_dist",
    "prefer_track_agent", "draw_map_resolution", "top_down_camera_initial_x", "top_down_camera_initial_y",
    "top_down_camera_initial_z", "window_size", "show_fps", "global_light", "onscreen_message", "debug_physics_world",
    "debug_static_world", "headless_machine_render", "pstats", "max_distance", "_disable_detector_mask",
    "load_map_from_json", "_load_map_from_json", "save_level",
    # unused upstream (no reader in the reference's reward function)
    "acceleration_penalty", "low_speed_penalty", "general_penalty",
}
VISUAL_VEHICLE_KEYS = {
    "show_navi_mark", "random_navi_mark_color", "show_dest_mark", "show_line_to_dest", "am_i_the_special_one", "show_lidar",
    "show_side_detector", "show_lane_line_detector", "mini_map", "rgb_camera", "depth_camera", "image_source", "random_color",
}
# NEUTRAL: features outside the step path (rendering, image observations, manual / scripted ego control, recording); the
# neutral value is accepted, anything else is refused by name instead of being silently ignored.
NEUTRAL_KEYS = {
    "use_render": False, "manual_control": False, "offscreen_render": False, "use_saver": False,
    "record_episode": False, "_debug_crash_object": False, "is_multi_agent": False, "num_agents": 1,
    "allow_respawn": False, "delay_done": 0, "gaussian_noise": 0.0, "dropout_prob": 0.0,
}
# overtake_stat: BaseVehicle._update_overtake_stat calls Lidar.get_surrounding_vehicles() without its argument in this
# version of the reference (base_vehicle.py:700-703, lidar.py:45-53): the switch cannot be turned on upstream either
NEUTRAL_VEHICLE_KEYS = {"extra_action_dim": 0, "overtake_stat": False}


def strip_reference_only_keys(user, keep=()):
    """Drops the VISUAL keys and the NEUTRAL keys at their neutral value; raises NotImplementedError for a NEUTRAL key
    that asks for a feature this engine does not have."""
    if not user:
        return user
    out = {}
    for k, v in user.items():
        if k in VISUAL_KEYS:
            continue
        if k in NEUTRAL_KEYS and k not in keep:
            if v != NEUTRAL_KEYS[k] and not (v is None and NEUTRAL_KEYS[k] is False):
                raise NotImplementedError("config['%s'] = %r: outside the step path this engine implements (only %r is "
                                          "accepted); see DESIGN.md section 8" % (k, v, NEUTRAL_KEYS[k]))
            continue
        if k == "vehicle_config" and isinstance(v, dict):
            vc = {}
            for kk, vv in v.items():
                if kk in VISUAL_VEHICLE_KEYS:
                    continue
                if kk in NEUTRAL_VEHICLE_KEYS:
                    if vv != NEUTRAL_VEHICLE_KEYS[kk]:
                        raise NotImplementedError("config['vehicle_config']['%s'] = %r is not supported (only %r)" % (
                            kk, vv, NEUTRAL_VEHICLE_KEYS[kk]))
                    continue
                vc[kk] = vv
            out[k] = vc
            continue
        out[k] = v
    return out


def resolve_map_choice(c, default_map=3):
    """parse_map_config (base_map.py:16-35): a `map_config` that names its own `config` wins (and `map` must then be left at its
    default, as upstream asserts); else `map` -- an int = number of blocks, a str = block sequence -- fills `type` / `config`.
    Returns the int or str and writes both forms back into the config."""
    mc = c["map_config"]
    if mc.get("config") is not None:
        if c["map"] != default_map and c["map"] != mc["config"]:
            raise ValueError("give the map either as `map` or as map_config['config'], not both (%r vs %r)" % (c["map"], mc["config"]))
        m = mc["config"]
        want = "block_num" if isinstance(m, int) else "block_sequence"
        if "type" in mc and mc["type"] not in (want, None) and not (mc["type"] == "block_num" and isinstance(m, str)):
            raise ValueError("map_config: type %r does not fit config %r" % (mc["type"], m))
    else:
        m = c["map"]
    if not isinstance(m, (int, str)) or isinstance(m, bool):
        raise ValueError("Unkown easy map config: %r" % (m, ))
    mc["type"] = "block_num" if isinstance(m, int) else "block_sequence"
    mc["config"] = m
    return m


def merge_config(default, user, path=""):
    """Nested update that rejects unknown keys (utils/config.py:115-125)."""
    out = copy.deepcopy(default)
    for k, v in (user or {}).items():
        if k not in out:
            raise KeyError("'{}' does not exist in existing config. Please use config.update(...) with known keys "
                           "only: {}".format(path + k, sorted(out.keys())))
        if isinstance(out[k], dict) and isinstance(v, dict):
            out[k] = merge_config(out[k], v, path + k + ".")
        elif isinstance(out[k], dict):  # a dict item overwritten by something that is not one (utils/config.py:143-152)
            raise TypeError("Type error! The item {} has original type {} and updating type {}.".format(path + k, type(out[k]), type(v)))
        else:
            out[k] = v
    return out


class PGDriveVecEnv:
    """N independent PGDrive environments stepped by one HIP launch pair per step."""
    def __init__(self, config=None):
        self.config = merge_config(DEFAULT_CONFIG, strip_reference_only_keys(config))
        c = self.config
        vc = c["vehicle_config"]
        # (the noise keys of the side / lane-line detectors exist upstream but are never read: only the lidar cloud is
        # perturbed, state_obs.py:155-170)
        mc = c["map_config"]
        seeds = list(range(c["start_seed"], c["start_seed"] + c["environment_num"]))
        if (c["random_lane_width"] or c["random_lane_num"]) and c["map_bank"] is not None:
            raise ValueError("random_lane_width / random_lane_num need generated maps: set map_bank=None "
                             "(upstream: 'You are supposed to turn off the load_map_from_json', map_manager.py:159-166)")
        if c["map_bank"] is not None:  # pre-generated descriptions (load_map_from_json, pgdrive_env.py:38-39)
            by_seed = {d["seed"]: d for d in bank.load_descriptions(c["map_bank"])}
            missing = [s for s in seeds if s not in by_seed]
            if missing:
                raise KeyError("map seeds %s..%s are not in the map bank" % (missing[0], missing[-1]))
        else:  # BIG on the host: `map` is a block count (int) or a block sequence (str) (base_map.py:16-35)
            m = resolve_map_choice(c)
            kw = dict(block_num=m) if isinstance(m, int) else dict(block_seq=m, block_num=None)
            by_seed = {d["seed"]: d for d in bank.get_descriptions(seeds, mc["lane_num"], mc["lane_width"],
                                                                   mc["exit_length"], random_lane_width=c["random_lane_width"],
                                                                   random_lane_num=c["random_lane_num"], **kw)}
        self.seeds = seeds
        sel = [by_seed[s] for s in seeds]
        self.num_envs = int(c["num_envs"])
        with_objects = abs(c["accident_prob"]) >= 1e-2
        T = int(c["max_traffic_vehicles"]) if abs(c["traffic_density"]) >= 1e-2 else 0
        T += int(c["max_traffic_objects"]) if with_objects else 0
        if 1 + T > 64:
            raise ValueError("at most 63 traffic + object slots per env")
        self.map_bank = mapdata.MapBank(sel)
        self.scen_bank = scenario.ScenarioBank(
            sel, seeds, num_agents=1, num_traffic=T, density=c["traffic_density"],
            spawn_longitude=vc["spawn_longitude"], spawn_lateral=vc["spawn_lateral"], vehicle_model=vc["vehicle_model"],
            spawn_lane_index=vc["spawn_lane_index"], destination_node=vc["destination_node"],
            traffic_mode=c["traffic_mode"], auto_termination=c["auto_termination"], accident_prob=c["accident_prob"],
            random_agent_model=c["random_agent_model"], idm_agent=bool(c["IDM_agent"]),
            traffic_seeds=np.random.RandomState(c["seed"]).randint(0, scenario.MAX_RAND_INT, len(seeds))
            if c["random_traffic"] else None
        )
        # the reference has no cap on traffic vehicles / objects: say so when the slot caps cut a scenario short
        lost = sum(i["dropped"] for i in self.scen_bank.info), sum(i["objects_dropped"] for i in self.scen_bank.info)
        if lost[0] or lost[1]:
            import warnings
            warnings.warn("pgdrive_amd: %d traffic vehicles and %d traffic objects of the reference's scenarios do not fit the slot "
                          "caps (max_traffic_vehicles=%d, max_traffic_objects=%d): raise the caps to reproduce the reference's "
                          "traffic" % (lost[0], lost[1], c["max_traffic_vehicles"], c["max_traffic_objects"]))
        lid, sd, ld = vc["lidar"], vc["side_detector"], vc["lane_line_detector"]
        nl = lid["num_lasers"] if lid["distance"] > 0 else 0
        self.cfg = _abi.make_config(
            self.num_envs, num_agents=1, num_traffic=T, num_lasers=nl, num_others=lid["num_others"],
            lidar_dist=lid["distance"], dt=c["physics_world_step_size"], decision_repeat=c["decision_repeat"],
            auto_reset=c["auto_reset"], resample_scenario=c["resample_scenario"], horizon=c["horizon"] or 0,
            seed=c["seed"], success_reward=c["success_reward"], out_of_road_penalty=c["out_of_road_penalty"],
            crash_vehicle_penalty=c["crash_vehicle_penalty"], crash_object_penalty=c["crash_object_penalty"],
            driving_reward=c["driving_reward"], speed_reward=c["speed_reward"], use_lateral=c["use_lateral"],
            out_of_route_done=c["out_of_route_done"],
            side_lasers=sd["num_lasers"] if sd["distance"] > 0 else 0, side_dist=sd["distance"],
            lane_line_lasers=ld["num_lasers"] if ld["distance"] > 0 else 0, lane_line_dist=ld["distance"],
            discrete_action=c["discrete_action"], discrete_steering_dim=c["discrete_steering_dim"],
            discrete_throttle_dim=c["discrete_throttle_dim"], increment_steering=vc["increment_steering"],
            safe_rl_env=c["safe_rl_env"], random_agent_model=c["random_agent_model"], enable_reverse=vc["enable_reverse"],
            lidar_gaussian_noise=lid["gaussian_noise"], lidar_dropout_prob=lid["dropout_prob"], idm_agent=bool(c["IDM_agent"]),
            idm_steer_lag=float(c["idm_steer_lag"])
        )
        from .engine import Engine
        self.engine = Engine(self.cfg, self.map_bank, self.scen_bank, device=c["device"])
        self.obs_dim = self.engine.D
        self._jit_thread = self.engine.specialise(wait=False) if c["jit_step_kernel"] else None  # (join() it to wait for the module)
        self.topdown = bool(c["use_topdown"])
        if self.topdown:
            # rgb_clip=False (pgdrive_env.py:133-141): the images as uint8 in [0, 255] (pgd_observe_topdown_u8) instead of float32 / 255
            # (the single-frame observation is built without a `resolution` argument upstream: TopDownObservation.RESOLUTION = 200,
            # top_down_env.py:23-26, top_down_obs.py:27)
            self.engine.enable_topdown(_abi.make_topdown_config(200 if c["topdown_single_frame"] else c["resolution_size"], c["distance"],
                                                                c["frame_stack"], c["post_stack"], c["frame_skip"],
                                                                mode=1 if c["topdown_single_frame"] else 0), uint8=not c["rgb_clip"])
        # spaces (base_vehicle.py:720-727, state_obs.py:124-130)
        self.single_observation_space = Box(-0.0, 1.0, (self.obs_dim, ), np.float32)
        if self.topdown:
            self.single_observation_space = Box(-0.0, 1.0, tuple(self.engine.img.shape[1:]), np.float32) if c["rgb_clip"] else \
                Box(0, 255, tuple(self.engine.img.shape[1:]), np.uint8)  # (top_down_obs_multi_channel.py:277-280)
        self.single_action_space = MultiDiscrete([c["discrete_steering_dim"], c["discrete_throttle_dim"]]) \
            if c["discrete_action"] else Box(-1.0, 1.0, (2, ), np.float32)
        self.observation_space = self.single_observation_space
        self.action_space = self.single_action_space
        self._rng = np.random.RandomState(c["seed"])

    def reset(self, force_seed=None):
        """Reset every env; `force_seed` (int or array) pins the map seed(s), else seeds are drawn uniformly from
        [start_seed, start_seed + environment_num) (base_env.py:451-458).  Returns a cuda float32 tensor [N, D].

        The tensors returned by reset() / step() are VIEWS of the engine's own output buffers, which the next step
        overwrites (no per-step allocation): a rollout buffer must copy what it keeps (`.clone()`), or pass its own buffers
        through `Engine.step(out=...)`."""
        if force_seed is None:
            ids = self._rng.randint(0, len(self.seeds), size=self.num_envs)
        else:
            fs = np.broadcast_to(np.asarray(force_seed), (self.num_envs, ))
            ids = np.array([self.seeds.index(int(s)) for s in fs])
        obs = self.engine.reset(ids.astype(np.int32))
        if self.topdown:
            return self.engine.observe_topdown()
        return obs.view(self.num_envs, self.obs_dim)

    def step(self, actions):
        """actions: cuda float32 tensor [N, 2] -> (obs [N,D], reward [N], done [N] uint8, flags [N] int32) on the GPU."""
        if self.config["vehicle_config"]["action_check"]:  # opt-in: costs a device round trip
            a = actions.reshape(self.num_envs, 2)
            if self.config["discrete_action"]:
                hi = a.new_tensor([self.config["discrete_steering_dim"] - 1, self.config["discrete_throttle_dim"] - 1])
                ok = bool(((a >= 0) & (a <= hi) & (a == a.round())).all())
            else:
                ok = bool(((a >= -1.0) & (a <= 1.0)).all())
            assert ok, "Input actions are not compatible with action space {}!".format(self.single_action_space)
        obs, rew, done, flags = self.engine.step(actions.contiguous().view(self.num_envs, 1, 2))
        if not getattr(self, "_kernel_reported", False):
            # once: a configuration that is not one of the reference's defaults runs the general step kernel, measured 12 % behind
            # the instantiations with the configuration compiled in -- say so instead of paying silently
            self._kernel_reported = True
            kname = self.engine.describe_step()
            if self.num_envs >= 1024 and "specialised" not in kname:
                import warnings
                warnings.warn("pgdrive_amd: this configuration runs the general step kernel (%s); the reference's default "
                              "configurations run specialised instantiations that are 12 - 17 %% faster -- jit_step_kernel=True (or "
                              "engine.specialise()) builds one for this configuration at run time" % kname)
        if self.topdown:
            return self.engine.observe_topdown(), rew.view(-1), done.view(-1), flags.view(-1)
        return obs.view(self.num_envs, self.obs_dim), rew.view(-1), done.view(-1), flags.view(-1)

    # -- asynchronous env groups (pgd_set_groups / pgd_step_group; double-buffered sampling: the policy of one group runs while the other
    # steps -- bench.py's closed-loop rows, examples/fused_policy_rollout.py).  Lidar observations only (the top-down image is one launch
    # over every env of the handle).
    def set_groups(self, n_groups):
        self.engine.set_groups(n_groups)

    def step_group(self, g, actions):
        """Step the envs of group g only, asynchronously on `engine.group_streams[g]`; `actions` is the full [N, 2] tensor.  Returns views
        of the group's rows (obs [n, D], reward, done, flags)."""
        if self.topdown:
            raise NotImplementedError("step_group with the top-down observation")
        obs, rew, done, flags = self.engine.step_group(g, actions.contiguous().view(self.num_envs, 1, 2))
        return obs.view(-1, self.obs_dim), rew.view(-1), done.view(-1), flags.view(-1)

    def group_sync(self, g):
        self.engine.group_sync(g)

    def group_slice(self, g):
        return self.engine.group_slice(g)

    def info_from_flags(self, flags):
        """Host-side decode of the flag bit-field into the reference's info keys (pgdrive_env.py:165-194)."""
        fl = np.asarray(flags.cpu() if hasattr(flags, "cpu") else flags).astype(np.uint32)
        return dict(
            arrive_dest=(fl & _abi.F_ARRIVE) != 0, out_of_road=(fl & _abi.F_OUT_OF_ROAD) != 0,
            crash_vehicle=(fl & _abi.F_CRASH_VEHICLE) != 0, crash_object=(fl & _abi.F_CRASH_OBJECT) != 0,
            crash_building=(fl & _abi.F_CRASH_BUILDING) != 0, max_step=(fl & _abi.F_MAX_STEP) != 0,
            crash=(fl & (_abi.F_CRASH_VEHICLE | _abi.F_CRASH_OBJECT | _abi.F_CRASH_BUILDING)) != 0,
        )

    def cost_from_flags(self, flags):
        """PGDriveEnv.cost_function (pgdrive_env.py:197-207): out_of_road, else crash_vehicle, else crash_object cost."""
        info = self.info_from_flags(flags)
        c = self.config
        return np.where(info["out_of_road"], c["out_of_road_cost"],
                        np.where(info["crash_vehicle"], c["crash_vehicle_cost"],
                                 np.where(info["crash_object"], c["crash_object_cost"], 0.0))).astype(np.float64)

    def seed(self, seed=None):
        if seed is not None:
            self._rng = np.random.RandomState(seed)

    def close(self):
        if getattr(self, "engine", None) is not None:
            self.engine.close()
            self.engine = None
