"""ctypes mirror of the plain-C structs in include/pgdrive_hip.h (kept in lock-step by tests/test_abi.py)."""
import ctypes as C


class PgdConfig(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("num_agents", C.c_int32), ("num_traffic", C.c_int32), ("num_lasers", C.c_int32),
        ("num_others", C.c_int32), ("lidar_dist", C.c_float), ("dt", C.c_float), ("decision_repeat", C.c_int32),
        ("auto_reset", C.c_int32), ("resample_scenario", C.c_int32), ("horizon", C.c_int32), ("seed", C.c_uint32),
        ("success_reward", C.c_float), ("out_of_road_penalty", C.c_float), ("crash_vehicle_penalty", C.c_float),
        ("crash_object_penalty", C.c_float), ("driving_reward", C.c_float), ("speed_reward", C.c_float),
        ("use_lateral", C.c_int32), ("out_of_route_done", C.c_int32), ("marl_flags", C.c_int32),
        ("delay_done", C.c_int32), ("agent_limit", C.c_int32), ("respawn_places", C.c_int32),
        ("respawn_dests", C.c_int32), ("side_lasers", C.c_int32), ("side_dist", C.c_float),
        ("lane_line_lasers", C.c_int32), ("lane_line_dist", C.c_float), ("discrete_action", C.c_int32),
        ("discrete_steering_dim", C.c_int32), ("discrete_throttle_dim", C.c_int32), ("increment_steering", C.c_int32),
        ("safe_rl_env", C.c_int32), ("overspeed_penalty", C.c_float), ("min_pass_steps", C.c_int32),
        ("enable_reverse", C.c_int32), ("lidar_gaussian_noise", C.c_float), ("lidar_dropout_prob", C.c_float),
        ("random_agent_model", C.c_int32), ("env_base", C.c_int32), ("idm_agent", C.c_int32),
        ("idm_steer_lag", C.c_float),
    ]


class TopDownConfig(C.Structure):
    """pgd_topdown_config; defaults = TopDownPGDriveEnv (envs/top_down_env.py:8-42)."""
    _fields_ = [("resolution", C.c_int32), ("distance", C.c_float), ("frame_stack", C.c_int32), ("post_stack", C.c_int32),
                ("frame_skip", C.c_int32), ("mode", C.c_int32)]


def make_topdown_config(resolution=84, distance=30.0, frame_stack=3, post_stack=5, frame_skip=5, mode=0):
    """mode 1: the single RGB frame of TopDownObservation (obs/top_down_obs.py; reference default resolution 200)."""
    return TopDownConfig(int(resolution), float(distance), int(frame_stack), int(post_stack), int(frame_skip), int(mode))


def make_config(num_envs, num_agents=1, num_traffic=16, num_lasers=240, num_others=4, lidar_dist=50.0, dt=0.02,
                decision_repeat=5, auto_reset=1, resample_scenario=0, horizon=0, seed=0, success_reward=10.0,
                out_of_road_penalty=5.0, crash_vehicle_penalty=5.0, crash_object_penalty=5.0, driving_reward=1.0,
                speed_reward=0.1, use_lateral=False, out_of_route_done=False, multi_agent=False, crash_done=True,
                out_of_road_done=True, allow_respawn=True, delay_done=25, agent_limit=0, respawn_places=0,
                respawn_dests=0, side_lasers=0, side_dist=50.0, lane_line_lasers=0, lane_line_dist=20.0,
                discrete_action=False, discrete_steering_dim=5, discrete_throttle_dim=5, increment_steering=False,
                safe_rl_env=False, plain_reward=False, cross_yellow_line_done=True, tollgate=False, overspeed_penalty=0.5,
                min_pass_steps=30, enable_reverse=False, parking=False, others_state=False, random_agent_model=False,
                lidar_gaussian_noise=0.0, lidar_dropout_prob=0.0, env_base=0, idm_agent=False, idm_steer_lag=0.0):
    """Defaults mirror PGDriveEnv_DEFAULT_CONFIG / BASE_DEFAULT_CONFIG (pgdrive_env.py:22-109, base_env.py:19-90)."""
    c = PgdConfig()
    c.num_envs, c.num_agents, c.num_traffic = num_envs, num_agents, num_traffic
    c.num_lasers, c.num_others, c.lidar_dist = num_lasers, (num_others if num_lasers > 0 else 0), lidar_dist
    c.dt, c.decision_repeat = dt, decision_repeat
    c.auto_reset, c.resample_scenario, c.horizon, c.seed = int(auto_reset), int(resample_scenario), int(horizon or 0), seed
    c.success_reward, c.out_of_road_penalty = success_reward, out_of_road_penalty
    c.crash_vehicle_penalty, c.crash_object_penalty = crash_vehicle_penalty, crash_object_penalty
    c.driving_reward, c.speed_reward = driving_reward, speed_reward
    c.use_lateral, c.out_of_route_done = int(bool(use_lateral)), int(bool(out_of_route_done))
    c.side_lasers, c.side_dist = int(side_lasers), float(side_dist)
    c.lane_line_lasers, c.lane_line_dist = int(lane_line_lasers), float(lane_line_dist)
    c.discrete_action, c.increment_steering = int(bool(discrete_action)), int(bool(increment_steering))
    c.discrete_steering_dim, c.discrete_throttle_dim = int(discrete_steering_dim), int(discrete_throttle_dim)
    c.safe_rl_env = int(bool(safe_rl_env))
    c.env_base = int(env_base)
    c.idm_agent = int(bool(idm_agent))
    c.idm_steer_lag = float(idm_steer_lag)  # (an extension, default off: include/pgdrive_hip.h)
    c.enable_reverse = int(bool(enable_reverse))
    c.random_agent_model = int(bool(random_agent_model))
    c.lidar_gaussian_noise, c.lidar_dropout_prob = float(lidar_gaussian_noise), float(lidar_dropout_prob)
    if multi_agent:
        c.marl_flags = MA_ENABLED | (MA_CRASH_DONE if crash_done else 0) | (MA_OUT_ROAD_DONE if out_of_road_done else 0) | \
            (MA_ALLOW_RESPAWN if allow_respawn else 0) | (MA_PLAIN_REWARD if plain_reward else 0) | \
            (0 if cross_yellow_line_done else MA_YELLOW_OK) | (MA_TOLLGATE if tollgate else 0) | \
            (MA_PARKING if parking else 0) | (MA_OTHERS_STATE if others_state else 0)
        c.overspeed_penalty, c.min_pass_steps = float(overspeed_penalty), int(min_pass_steps)
        c.delay_done, c.agent_limit = int(delay_done), int(agent_limit or num_agents)
        c.respawn_places, c.respawn_dests = int(respawn_places), int(respawn_dests)
    return c


def obs_dim(cfg):
    toll = bool(cfg.marl_flags & MA_TOLLGATE)
    state = (cfg.side_lasers or 2) + 6 + cfg.lane_line_lasers + (2 if cfg.random_agent_model else 0) + (0 if toll else 10)
    per_other = state if cfg.marl_flags & MA_OTHERS_STATE else 4
    return state + per_other * cfg.num_others + cfg.num_lasers + (2 if toll else 0)


# state layout (include/pgd_state_layout.h)
SF = dict(X=0, Y=1, THETA=2, SPEED=3, STEER=4, THROTTLE=5, LASTX=6, LASTY=7, LASTHX=8, LASTHY=9, ACT0S=10, ACT0T=11,
          ACT1S=12, ACT1T=13, PID_HP=14, PID_HI=15, PID_LP=16, PID_LI=17, TARGET_SPEED=18, ENERGY=19, DIST_LEFT=20,
          DIST_RIGHT=21, EP_REWARD=22, AGENT_ID=23, HX=24, HY=25)
SI = dict(STATUS=0, LANE=1, CK0=2, CK1=3, RLANE=4, TIMER=5, VFLAGS=6, SPAWN=7)
EI = dict(SCEN=0, NEXT_GROUP=1, EP_STEPS=2, EPISODES=3, STEPS_TOTAL=4, NEXT_AGENT=5, AUX=6, NEAR=7)
NF, NI, NEI = 26, 8, 8
ST_EMPTY, ST_PENDING, ST_ACTIVE, ST_REMOVED, ST_DYING = 0, 1, 2, 3, 4
MA_ENABLED, MA_CRASH_DONE, MA_OUT_ROAD_DONE, MA_ALLOW_RESPAWN, MA_PLAIN_REWARD, MA_YELLOW_OK, MA_TOLLGATE = 1, 2, 4, 8, 16, 32, 64
MA_PARKING = 128
MA_OTHERS_STATE = 256

F_ARRIVE, F_OUT_OF_ROAD, F_CRASH_VEHICLE, F_CRASH_OBJECT, F_CRASH_BUILDING, F_MAX_STEP = 1, 2, 4, 8, 16, 32
F_ON_YELLOW, F_ON_WHITE, F_ON_BROKEN, F_CRASH_SIDEWALK, F_OFF_LANE, F_OUT_OF_ROUTE = 256, 512, 1024, 2048, 4096, 8192
F_OBJECT_HIT = 1 << 14
F_RESET, F_REPORT, F_NEW, F_ALL_DONE = 1 << 16, 1 << 17, 1 << 18, 1 << 19
