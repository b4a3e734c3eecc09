"""Single-environment, numpy-in / numpy-out wrapper with the reference's exact gym.Env call shapes.

    env = PGDriveEnv(dict(start_seed=1000, environment_num=100))
    o = env.reset(); o, r, d, info = env.step([0.0, 1.0])

Mirrors pgdrive/envs/pgdrive_env.py (PGDriveEnv) on top of the batched engine with N = 1 and auto_reset off, so the
terminal observation is returned and the user calls reset() — exactly the reference's episode protocol
(envs/base_env.py:184-193, 269-344).  Gym ids -> seed ranges follow pgdrive/register.py:5-38.
"""
import numpy as np

from . import _abi
from .spaces import EnvBase
from .vec_env import PGDriveVecEnv

# register.py:5-38
ENV_IDS = {
    "PGDrive-test-v0": dict(start_seed=0, environment_num=200),
    "PGDrive-validation-v0": dict(start_seed=200, environment_num=800),
    "PGDrive-v0": dict(start_seed=1000, environment_num=100),
    "PGDrive-10envs-v0": dict(start_seed=1000, environment_num=10),
    "PGDrive-1000envs-v0": dict(start_seed=1000, environment_num=1000),
    "PGDrive-training0-v0": dict(start_seed=3000, environment_num=1000),
    "PGDrive-training1-v0": dict(start_seed=5000, environment_num=1000),
    "PGDrive-training2-v0": dict(start_seed=7000, environment_num=1000),
}


class PGDriveEnv(EnvBase):  # gym.Env when gym is importable (base_env.py:93), else object: pgdrive_amd/spaces.py
    metadata = {"render.modes": []}

    def __init__(self, config=None):
        cfg = dict(config or {})
        cfg["num_envs"] = 1
        # the step info (step_reward, velocity, energy) is read from the engine state AFTER the step: with an in-kernel
        # auto-reset that state would already be the next episode's on exactly the terminal steps.  The single-env surface
        # resets through reset(), like the reference's gym.Env.
        if cfg.get("auto_reset") or cfg.get("resample_scenario"):
            raise ValueError("PGDriveEnv resets through reset(): auto_reset / resample_scenario belong to PGDriveVecEnv")
        cfg["auto_reset"] = False
        cfg["resample_scenario"] = False
        self.vec = PGDriveVecEnv(cfg)
        self.config = self.vec.config
        self.observation_space = self.vec.single_observation_space
        self.action_space = self.vec.single_action_space
        self.episode_steps = 0
        self.episode_reward = 0.0
        self._done = False
        self._last_energy = 0.0
        import torch
        self._torch = torch

    def reset(self, episode_data=None, force_seed=None):
        assert episode_data is None, "episode replay is not built"
        obs = self.vec.reset(force_seed=force_seed)
        self.episode_steps = 0
        self.episode_reward = 0.0
        self._done = False
        self._last_energy = 0.0
        return obs[0].cpu().numpy()

    def step(self, action):
        a = self._torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(1, 2), device=self.vec.engine.device)
        obs, rew, done, flags = self.vec.step(a)
        self.vec.engine.sync()
        self.episode_steps += 1
        r = float(rew[0].item())
        d = bool(done[0].item()) or self._done  # sticky done (base_env.py:318-319)
        self._done = d
        self.episode_reward += r
        fl = int(flags[0].item()) & 0xFFFFFFFF
        f, i, ei = self.vec.engine.get_state()
        info = {k: bool(np.asarray(v).reshape(-1)[0]) for k, v in self.vec.info_from_flags(np.array([fl])).items()}
        cost = float(self.vec.cost_from_flags(np.array([fl]))[0])  # cost_function (pgdrive_env.py:197-207)
        energy = float(f[_abi.SF["ENERGY"], 0, 0])
        # step_reward is the shaping reward BEFORE the terminal override (pgdrive_env.py:236-246); a terminal step's is
        # re-derived on the host from the state the engine left (single env: a few lane formulas)
        shaping = r if not (info["arrive_dest"] or info["out_of_road"] or info["crash_vehicle"] or info["crash_object"]) \
            else self._shaping_reward(f, i, ei)
        raw = np.asarray(action, dtype=np.float64).reshape(-1)
        # the keys of BaseVehicle.after_step (base_vehicle.py:255-273), _preprocess_action (:231-236), reward / cost / done
        # functions (pgdrive_env.py:162-258) and _get_step_return (base_env.py:303-344)
        info.update(
            cost=cost, velocity=abs(float(f[_abi.SF["SPEED"], 0, 0])) * 3.6, steering=float(f[_abi.SF["STEER"], 0, 0]),
            acceleration=float(f[_abi.SF["THROTTLE"], 0, 0]), step_reward=shaping, episode_reward=self.episode_reward,
            episode_length=self.episode_steps, episode_energy=energy, step_energy=energy - self._last_energy,
            raw_action=(float(raw[0]), float(raw[1])), overtake_vehicle_num=0,  # overtake_stat cannot be on (vec_env.py)
        )
        self._last_energy = energy
        return obs[0].cpu().numpy(), r, d, info

    def _shaping_reward(self, f, i, ei):
        """PGDriveEnv.reward_function up to `step_info["step_reward"] = reward` (pgdrive_env.py:209-236) from the state."""
        from . import mapdata
        SF, SI, c = _abi.SF, _abi.SI, self.config
        scen = int(ei[_abi.EI["SCEN"], 0])
        sb, mb = self.vec.scen_bank, self.vec.map_bank
        d = mb.descs[int(sb.scenarios["map"][scen])]
        sp = sb.spawns[scen * sb.V]
        lane = d["lanes"][int(i[SI["LANE"], 0, 0])]
        cur_road = int(sp["ckpt_road"][int(i[SI["CK0"], 0, 0])])
        road = d["roads"][cur_road]
        positive = 1.0
        if lane["road"] != cur_road:  # off the reference lanes: first reference lane, sign of the road the car is on
            positive = -1.0 if d["roads"][lane["road"]]["negative"] else 1.0
            lane = d["lanes"][road["first_lane"]]
        l0, _ = mapdata.lane_local_coordinates(lane, (float(f[SF["LASTX"], 0, 0]), float(f[SF["LASTY"], 0, 0])))
        l1, t1 = mapdata.lane_local_coordinates(lane, (float(f[SF["X"], 0, 0]), float(f[SF["Y"], 0, 0])))
        lateral = min(max(1.0 - 2.0 * abs(t1) / d["lane_width"], 0.0), 1.0) if c["use_lateral"] else 1.0
        speed_kmh = abs(float(f[SF["SPEED"], 0, 0])) * 3.6
        return c["driving_reward"] * (l1 - l0) * lateral * positive + c["speed_reward"] * (speed_kmh / float(sp["max_speed"])) * positive

    def seed(self, seed=None):
        self.vec.seed(seed)

    def close(self):
        self.vec.close()


class SafePGDriveEnv(PGDriveEnv):
    """pgdrive/envs/safe_pgdrive_env.py:7-60: accident scenes (traffic cones, broken-down vehicles with warning tripods,
    barriers; manager/object_manager.py) on the Straight / Curve / ramp blocks, crashes are costs instead of
    terminations, `info["total_cost"]` accumulates over the episode."""
    DEFAULTS = dict(environment_num=100, accident_prob=0.8, traffic_density=0.05, safe_rl_env=True, crash_vehicle_cost=1.0,
                    crash_object_cost=1.0, out_of_road_cost=1.0, use_lateral=False)

    def __init__(self, config=None):
        cfg = dict(self.DEFAULTS)
        user = dict(config or {})
        cost_to_reward = user.pop("cost_to_reward", False)
        cfg.update(user)
        if cost_to_reward:  # _post_process_config (safe_pgdrive_env.py:41-47)
            from .vec_env import DEFAULT_CONFIG
            for pen, cost in (("crash_vehicle_penalty", "crash_vehicle_cost"), ("crash_object_penalty", "crash_object_cost"),
                              ("out_of_road_penalty", "out_of_road_cost")):
                cfg[pen] = cfg.get(pen, DEFAULT_CONFIG[pen]) + cfg[cost]
        super().__init__(cfg)
        self.episode_cost = 0.0

    def reset(self, *args, **kwargs):
        self.episode_cost = 0.0
        return super().reset(*args, **kwargs)

    def step(self, action):
        o, r, d, info = super().step(action)
        self.episode_cost += info["cost"]
        info["total_cost"] = self.episode_cost
        return o, r, d, info


class TopDownPGDriveEnv(PGDriveEnv):
    """pgdrive/envs/top_down_env.py:28-42: PGDriveEnv with the TopDownMultiChannel observation, [84, 84, 2 + frame_stack]
    float32 in [0, 1], the lidar switched off."""
    DEFAULTS = dict(use_topdown=True, frame_skip=5, frame_stack=3, post_stack=5, rgb_clip=True, resolution_size=84, distance=30)

    def __init__(self, config=None):
        cfg = dict(self.DEFAULTS)
        user = dict(config or {})
        vc = dict(user.pop("vehicle_config", {}) or {})
        vc.setdefault("lidar", dict(num_lasers=0, distance=0))  # "Remove lidar" (top_down_env.py:12)
        cfg.update(user)
        cfg["vehicle_config"] = vc
        super().__init__(cfg)


class TopDownPGDriveEnvV2(TopDownPGDriveEnv):
    """pgdrive/envs/top_down_env.py:44-72: the same observation and defaults as TopDownPGDriveEnv, derived from PGDriveEnv directly
    upstream (its lidar entry is replaced instead of updated: the remaining lidar keys do not reach the observation)."""


class TopDownSingleFramePGDriveEnv(TopDownPGDriveEnv):
    """pgdrive/envs/top_down_env.py:8-26: PGDriveEnv with TopDownObservation (obs/top_down_obs.py): ONE RGB frame [200, 200, 3]
    float32 in [0, 1] -- lane lines (35, 35, 35), the ego green (50, 200, 0), the other vehicles blue (100, 200, 255), +-30 m
    around the ego, ego heading up; frame_stack / post_stack / frame_skip are config keys upstream that this observation
    never reads.  What is drawn exactly: pgdrive_amd/csrc/pgd_topdown.h (pygame's rasterisation is unpinned)."""
    DEFAULTS = dict(TopDownPGDriveEnv.DEFAULTS, topdown_single_frame=True)


def make(env_id, **kw):
    cfg = dict(ENV_IDS[env_id])
    cfg.update(kw)
    return PGDriveEnv(cfg)
