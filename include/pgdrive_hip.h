/*
 * pgdrive_hip.h — C ABI of the MI355X-native batched PGDrive step engine.
 *
 * The reference (decisionforce/pgdrive v0.1.4) has no FFI/plugin layer for env.step(); its boundary is the Python
 * gym.Env surface (pgdrive/envs/base_env.py:184-193 step, :269-290 reset, :403-425 spaces).  This header re-states that
 * surface batched over N environments x V vehicle slots behind plain C entry points (no torch / C++ types), so a Python
 * VecEnv (pgdrive_amd/vec_env.py, ctypes) or any other host can bind it.  Each entry point cites what it replaces.
 *
 * Conventions: int status codes (0 = PGD_OK); the caller owns every buffer; pointers named d_* are DEVICE pointers,
 * h_* are HOST pointers; all work is enqueued on the hipStream_t given at creation (passed as void* so this header
 * needs no HIP include); no hidden global state; one handle may be driven by one host thread at a time.
 *
 * Geometry contract (all float32, PGDrive coordinates: x forward, y left->right as in the reference, angles in rad):
 *   pgd_lane  : StraightLane / CircularLane closed forms  (component/lane/straight_lane.py:13-67, circular_lane.py:11-67)
 *   pgd_road  : RoadNetwork.graph[from][to] -> lanes     (component/road/road_network.py:20)
 *   pgd_box   : every box the reference hands to Bullet   (component/blocks/base_block.py:286-464): lane-surface boxes
 *               (kind 0), white/yellow continuous and broken line ghosts (1,2,3), sidewalk bodies (4)
 *   uniform grid per map: cell -> ascending list of box ids (creation order == the reference's Bullet insertion order)
 */
#ifndef PGDRIVE_HIP_H
#define PGDRIVE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGD_OK 0
#define PGD_ERR_ARG 1
#define PGD_ERR_HIP 2
#define PGD_ERR_STATE 3

#define PGD_MAX_SUCC 8       /* successor lanes stored inline per lane */
#define PGD_MAX_CKPT 32      /* max nodes of a route (Navigation.checkpoints, navigation.py:131) */
#define PGD_NAVI_DIM 10      /* Navigation.navigation_info_dim (navigation.py:23) */
#define PGD_STATE_DIM 8      /* ego floats of StateObservation.vehicle_state at default config (state_obs.py:58-106) */

/* box kinds (constants.py:55-63 BodyName) */
#define PGD_BOX_LANE 0
#define PGD_BOX_WHITE 1
#define PGD_BOX_YELLOW 2
#define PGD_BOX_BROKEN 3
#define PGD_BOX_SIDEWALK 4

/* per-agent flag bits returned by pgd_step (constants.py:15-22 TerminationState + BaseVehicleState) */
#define PGD_F_ARRIVE        (1u << 0)   /* arrive_dest          base_vehicle.py:738-745 */
#define PGD_F_OUT_OF_ROAD   (1u << 1)   /* out_of_road          pgdrive_env.py:209-216 */
#define PGD_F_CRASH_VEHICLE (1u << 2)   /* crash_vehicle        collision_callback.py:7-36 */
#define PGD_F_CRASH_OBJECT  (1u << 3)   /* crash_object: first contact with a traffic object (collision_callback.py:27-32) */
#define PGD_F_CRASH_BUILDING (1u << 4)  /* crash_building (InvisibleWall/TollGate: always 0 on PG maps) */
#define PGD_F_MAX_STEP      (1u << 5)   /* max_step / horizon   base_env.py:190-192 */
#define PGD_F_ON_YELLOW     (1u << 8)   /* on_yellow_continuous_line  base_vehicle.py:615-636 */
#define PGD_F_ON_WHITE      (1u << 9)   /* on_white_continuous_line */
#define PGD_F_ON_BROKEN     (1u << 10)  /* on_broken_line */
#define PGD_F_CRASH_SIDEWALK (1u << 11) /* crash_sidewalk        base_vehicle.py:638-643 */
#define PGD_F_OFF_LANE      (1u << 12)  /* not on_lane           navigation.py:158-160 */
#define PGD_F_OUT_OF_ROUTE  (1u << 13)  /* out_of_route          base_vehicle.py:274-276 */
#define PGD_F_OBJECT_HIT    (1u << 14)  /* state bit of a traffic OBJECT's own slot: TrafficObject.crashed (COST_ONCE,
                                           collision_callback.py:28-32); never reported for agents */
#define PGD_F_RESET         (1u << 16)  /* this env was auto-reset at the end of the step (obs is the new episode's) */
#define PGD_F_REPORT        (1u << 17)  /* multi-agent: the slot held an active agent this step: obs/reward/done are valid */
#define PGD_F_NEW           (1u << 18)  /* multi-agent: an agent was (re)spawned into this slot: obs valid, reward 0 */
#define PGD_F_ALL_DONE      (1u << 19)  /* multi-agent: done["__all__"] (multi_agent_pgdrive.py:142-148) */

typedef struct __attribute__((aligned(16))) pgd_lane {   /* 64 B */
  float ax, ay;             /* straight: start point; circular: centre */
  float bx, by;             /* straight: unit direction; circular: (radius, start_phase) */
  float c;                  /* straight: heading; circular: end_phase */
  float dir;                /* 0 = straight; +1/-1 = CircularLane.direction */
  float length, width;
  float ex, ey;             /* lane end point = position(length, 0) */
  int16_t road;             /* map-local road id */
  int16_t index;            /* lane index inside its road (0 = left-most) */
  int16_t n_succ;
  int16_t pad;
  int16_t succ[PGD_MAX_SUCC]; /* map-local lane ids L2 with |end - L2.start| < 0.1 (abs_lane.py:114-119) */
} pgd_lane;

typedef struct __attribute__((aligned(16))) pgd_road {   /* 16 B */
  int16_t from, to;         /* node ids */
  int16_t first_lane, n_lanes;
  uint8_t negative;         /* Road.is_negative_road (road.py:33-34) */
  uint8_t block_id;         /* Road.block_ID char (road.py:46-51) */
  uint8_t valid;            /* 0 for the decoration road */
  uint8_t pad0;
  int32_t pad1;
} pgd_road;

typedef struct __attribute__((aligned(16))) pgd_box {    /* 32 B */
  float cx, cy;             /* centre */
  float ux, uy;             /* unit vector of the long axis */
  float hl, hw;             /* half extents along / across */
  int32_t kind;             /* PGD_BOX_* */
  int32_t lane;             /* map-local lane id for PGD_BOX_LANE, else -1 */
} pgd_box;

typedef struct pgd_map {    /* per-map header; offsets index the bank-wide arrays */
  int32_t lane_off, n_lanes;
  int32_t road_off, n_roads;
  int32_t box_off, n_boxes;
  int32_t cell_off;         /* into cell_start[] (gx*gy+1 entries for this map) */
  int32_t item_off;         /* into cell_items[] */
  int32_t gx, gy;
  float ox, oy, cell;       /* grid origin and cell size [m] */
  float lane_width;         /* map config lane_width (Navigation.get_current_lane_width, navigation.py:322-323) */
  int32_t pad[2];
} pgd_map;

/* One vehicle slot of a scenario (spawn state + static parameters).  Slots [0,A) are controlled agents, the rest are
 * IDM traffic (manager/traffic_manager.py:239-290).  A slot with lane < 0 is unused. */
typedef struct pgd_spawn {  /* 64 B + route */
  float x, y, heading;      /* spawn pose: lane.position(long, lat), lane.heading_at(long)  (base_vehicle.py:304-309) */
  float length, width;      /* vehicle_type.py:7-74 */
  float wheelbase;          /* FRONT_WHEELBASE + REAR_WHEELBASE */
  float mass;
  float max_engine_force, max_brake_force, friction; /* utils/space.py:219-255 */
  float max_steer;          /* [rad] */
  float max_speed;          /* [km/h] */
  int16_t lane;             /* spawn lane (map-local); <0 = empty slot */
  int16_t group;            /* trigger group (block order); -1 = active from the start (agents) */
  int16_t n_ckpt;           /* route length in nodes */
  int16_t timer0;           /* IDMPolicy.overtake_timer initial value (idm_policy.py:185) */
  int16_t dest_lane;        /* Navigation.final_lane (navigation.py:138-140) */
  int16_t kind;             /* PGD_OBJ_*: what occupies the slot */
  int16_t aux;              /* PGD_MA_PARKING: 1 + index of the parking space this vehicle drives to (0 = none) */
  int16_t pad;
  int16_t ckpt[PGD_MAX_CKPT];       /* route as node ids (Navigation.checkpoints) */
  int16_t ckpt_road[PGD_MAX_CKPT];  /* road id of (ckpt[k], ckpt[k+1]); -1 past the end */
} pgd_spawn;

/* pgd_spawn.kind.  Traffic objects (manager/object_manager.py, static_object/traffic_object.py) and broken-down vehicles
 * occupy traffic slots with group = PGD_GROUP_NEVER: present in the world (contacts, lidar, IDM neighbour search), never
 * driven.  Deliberate substitution: Bullet lets a hit cone fly away; here objects are static. */
#define PGD_OBJ_VEHICLE  0  /* chassis box length x width */
#define PGD_OBJ_CYLINDER 1  /* TrafficCone / TrafficWarning: circle of radius length / 2 (traffic_object.py:40,60) */
#define PGD_OBJ_BOX      2  /* TrafficBarrier: box length x width at `heading` (traffic_object.py:80-96) */
#define PGD_OBJ_BUILDING 3  /* TollGateBuilding: invisible wall length x width (tollgate_building.py, scene_utils.py:260-295);
                               a contact is crash_building (collision_callback.py:33-35), every step */
#define PGD_GROUP_NEVER (-2)

typedef struct pgd_scenario {
  int32_t map;              /* index into the map table */
  int32_t n_groups;         /* number of traffic trigger groups */
  int16_t trigger_road[16]; /* BlockVehicles.trigger_road per group, in activation order (traffic_manager.py:283-288) */
  int32_t max_steps;        /* auto_termination: 250 * map.num_blocks (base_env.py:318); 0 = off */
  int32_t aux;              /* PGD_MA_PARKING: bit k = parking space k is nobody's destination at reset */
} pgd_scenario;

typedef struct pgd_config {
  int32_t num_envs;         /* N */
  int32_t num_agents;       /* A  controlled agents per env (1 = PGDriveEnv; >1 = MARL) */
  int32_t num_traffic;      /* T  IDM traffic slots per env; V = A + T */
  int32_t num_lasers;       /* lidar beams (pgdrive_env.py:63; 0 disables the lidar block of the obs) */
  int32_t num_others;       /* neighbour-info vehicles (lidar.num_others, 4) */
  float lidar_dist;         /* 50 m */
  float dt;                 /* physics_world_step_size 0.02 (base_env.py:70) */
  int32_t decision_repeat;  /* 5 (base_env.py:33) */
  int32_t auto_reset;       /* 1: envs whose agent is done are reset inside pgd_step */
  int32_t resample_scenario;/* 1: on auto-reset draw a new scenario (base_env.py:451-458), else keep env's scenario */
  int32_t horizon;          /* 0 = none (base_env.py:190-192) */
  uint32_t seed;            /* seed of the device-side counter RNG used for scenario resampling */
  /* reward scheme (pgdrive_env.py:91-101) */
  float success_reward, out_of_road_penalty, crash_vehicle_penalty, crash_object_penalty;
  float driving_reward, speed_reward;
  int32_t use_lateral;
  int32_t out_of_route_done;
  /* multi-agent (envs/marl_envs/multi_agent_pgdrive.py:12-55); all zero for the single-agent PGDriveEnv */
  int32_t marl_flags;       /* PGD_MA_* bits */
  int32_t delay_done;       /* steps a finished agent stays as a static obstacle (25) */
  int32_t agent_limit;      /* num_agents of the reference: respawn only while active + dying < agent_limit */
  int32_t respawn_places;   /* P safe spawn places (SpawnManager.safe_spawn_places, spawn_manager.py:114-155) */
  int32_t respawn_dests;    /* Dn destinations; every scenario carries P*Dn extra pgd_spawn records after its V slots */
  /* SideDetector / LaneLineDetector ray fans against lane-line boxes (vehicle_module/distance_detector.py:137-152);
   * 0 lasers (the reference default, pgdrive_env.py:64-65) keeps the two lateral-distance floats of state_obs.py:66-71 */
  int32_t side_lasers;      /* k: rays vs continuous (white / yellow / side) line boxes; replace obs[0:2] when > 0 */
  float side_dist;          /* 50 m */
  int32_t lane_line_lasers; /* m: rays vs continuous + broken line boxes; inserted after the yaw-rate float when > 0 */
  float lane_line_dist;     /* 20 m */
  /* action pre-processing of the controlled agents */
  int32_t discrete_action;  /* 1: EnvInputPolicy.convert_to_continuous_action (env_input_policy.py:28-31), applied AFTER the
                               clip to [-1,1] exactly as the reference does */
  int32_t discrete_steering_dim, discrete_throttle_dim; /* 5, 5 (base_env.py:35-36) */
  int32_t increment_steering; /* 1: steering += a0 * 0.05, clipped (base_vehicle.py:351-358) */
  int32_t safe_rl_env;      /* 1: SafePGDriveEnv.done_function (safe_pgdrive_env.py:49-56): a step with crash_vehicle, else
                               crash_object, is never terminal */
  /* MultiAgentTollgateEnv (envs/marl_envs/marl_tollgate.py), read when PGD_MA_TOLLGATE is set */
  float overspeed_penalty;  /* 0.5: reward = -penalty * speed / max_speed while too fast inside the toll block */
  int32_t min_pass_steps;   /* 30: an agent that crossed the toll block in fewer steps is terminated (out_of_road) */
  int32_t enable_reverse;   /* vehicle_config.enable_reverse (base_vehicle.py:366-376) for the controlled agents: negative
                               throttle drives backwards instead of braking (parking-lot env) */
  /* LidarStateObservation._add_noise_to_cloud_points (state_obs.py:172-182): beams <- clip(beam + N(0, sigma), 0, 1), then
   * 0 with probability dropout.  The reference draws from the global numpy RNG; here a counter-based stream of
   * (seed, env, agent, beam, step) -- same distribution, reproducible */
  float lidar_gaussian_noise, lidar_dropout_prob;
  int32_t random_agent_model; /* 1: two more state floats, LENGTH / 10 and WIDTH / 2.5, after the lane-line fan
                               (state_obs.py:21-22,102-105); the vehicle type itself comes with the spawn record */
  int32_t env_base;         /* global index of this engine's env 0.  The device RNG streams (IDM timers, lidar noise, scenario
                               re-draws, respawn destinations) are keyed by env_base + e, so envs sharded over several engines
                               / GPUs reproduce the single-engine run env for env */
  int32_t idm_agent;        /* IDM_agent (base_env.py:30, agent_manager.py:79): the ego is driven by IDMPolicy along its route,
                               the actions handed to pgd_step are ignored.  Single-agent engines only */
  float idm_steer_lag;      /* NOT in the reference; 0 (the default) = the reference's behaviour.  > 0: time constant [s] of a
                               first-order lag between the steering IDMPolicy commands (idm_policy.py:244-252) and the steering an
                               IDM-driven vehicle applies: steer += (clip(cmd) - steer) * T / (lag + T), T = dt * decision_repeat.
                               The reference's heading PID (kp 1.7, kd 3.5 per 0.1 s decision) was tuned on Bullet's raycast
                               vehicle; on the kinematic bicycle (no yaw inertia) the same gains end in a two-step limit cycle,
                               steering lock to lock (tests/test_traffic_band_gpu.py).  The lag stands in for the yaw dynamics the
                               bicycle lacks: an OPT-IN for users who train against traffic; unpinned like the bicycle itself.
                               Controlled agents are never lagged.  0.2 s settles the traffic on its lane axis */
} pgd_config;

#define PGD_MA_ENABLED        1  /* MultiAgentPGDrive semantics: per-agent done, delay-done queue, respawn, __all__ */
#define PGD_MA_CRASH_DONE     2  /* crash_done       (multi_agent_pgdrive.py:21) */
#define PGD_MA_OUT_ROAD_DONE  4  /* out_of_road_done (multi_agent_pgdrive.py:22) */
#define PGD_MA_ALLOW_RESPAWN  8  /* allow_respawn    (multi_agent_pgdrive.py:26) */
#define PGD_MA_PLAIN_REWARD  16  /* MultiAgentBottleneckEnv.reward_function (marl_bottleneck.py:91-128): no -1 factor on a
                                    negative road when the vehicle is off its reference lanes */
#define PGD_MA_TOLLGATE      64  /* MultiAgentTollgateEnv: toll reward / out-of-road / stay-time rules, observation without the
                                    navigation block plus 2 toll floats (marl_tollgate.py:63-105,195-270) */
#define PGD_MA_PARKING      128  /* MultiAgentParkingLotEnv (marl_parking_lot.py:39-90,160-222): destinations of agents entering
                                    from a road are parking spaces handed out from a per-env pool (released when the agent
                                    is done); a road place is respawned into only while a space is free; out-of-road =
                                    yellow line / off lane / sidewalk (white lines may be crossed) */
#define PGD_MA_YELLOW_OK     32  /* cross_yellow_line_done = False (marl_bottleneck.py:130-136): a yellow line is not
                                    out-of-road */
#define PGD_MA_OTHERS_STATE 256  /* LidarStateObservationMARound.observe (marl_inout_roundabout.py:82-105): each of the
                                    num_others neighbour rows is the neighbour's own StateObservation vector (as long as
                                    the observing agent's state block, zeros when absent) instead of 4 relative floats */

typedef struct pgd_engine* pgd_handle;

/* Size of one observation row D = (side_lasers or 2) + 6 + lane_line_lasers [+ 2 if random_agent_model] + 10 + 4*num_others + num_lasers
 * (obs/state_obs.py:17-23,108-114,124-130); 274 at the defaults.  With PGD_MA_TOLLGATE the 10 navigation floats are
 * absent and 2 toll floats follow the lidar (marl_tollgate.py:63-105). */
int pgd_obs_dim(const pgd_config* cfg);

/* Replaces PGDriveEnv.__init__ / lazy_init (envs/base_env.py:100-178): allocates device state for N x V slots. */
int pgd_create(const pgd_config* cfg, int device, void* hip_stream, pgd_handle* out);

/* Replaces MapManager.update_map + block._create_in_world (manager/map_manager.py:98-155, blocks/base_block.py:142-179):
 * uploads immutable geometry for a bank of maps.  All pointers are HOST arrays; they are copied. */
int pgd_upload_maps(pgd_handle h, const pgd_map* h_maps, int n_maps, const pgd_lane* h_lanes, int n_lanes,
                    const pgd_road* h_roads, int n_roads, const pgd_box* h_boxes, int n_boxes,
                    const int32_t* h_cell_start, int n_cell_start, const int32_t* h_cell_items, int n_cell_items);

/* Replaces AgentManager.reset + TrafficManager.reset spawn tables (manager/agent_manager.py:91-132,
 * traffic_manager.py:48-69,239-290): a bank of scenarios, each with V spawn slots.  HOST arrays, copied. */
int pgd_upload_scenarios(pgd_handle h, const pgd_scenario* h_scen, int n_scen, const pgd_spawn* h_spawns /*[n_scen*V]*/);

/* Replaces env.reset(force_seed) (envs/base_env.py:269-301) for the listed envs (h_env_ids == NULL -> all):
 * env e starts scenario h_scen_ids[i]; writes the first observation.  d_obs may be NULL. */
int pgd_reset(pgd_handle h, const int32_t* h_env_ids, const int32_t* h_scen_ids, int n, float* d_obs /*[N,A,D]*/);

/* Replaces env.step(action) (envs/base_env.py:184-224, 303-344) for all N envs.  Asynchronous on the stream.
 * Multi-agent engines: the row of a slot that is not due in this step (no agent, or an agent that did not report) reads zero.  The
 * engine writes such a row once and remembers that it did -- per env, together with the identity of the buffer (address and row
 * stride) the marks describe; the kernel compares that identity itself, so eager calls with alternating buffers, HIP-graph replays
 * and env groups on their own streams all see marks that belong to the buffer they write (the same rule for pgd_reset /
 * pgd_observe / pgd_step_packed).  A caller that scribbles over rows it was handed (in-place normalisation, noise) must not expect
 * them to be zeroed again while it keeps passing the same buffer: copy first, pass another buffer, or create the engine with
 * PGD_NO_ROWZ=1 in the environment (every row that is not due is then zero-filled by every call). */
int pgd_step(pgd_handle h, const float* d_actions /*[N,A,2]*/, float* d_obs /*[N,A,D]*/, float* d_reward /*[N,A]*/,
             uint8_t* d_done /*[N,A]*/, uint32_t* d_flags /*[N,A]*/);

/* K steps of an action ring in one call (open-loop use: action repeat / frame skip, scripted roll-outs, benchmarks): step k
 * (k = 0 .. n_steps - 1) applies d_action_ring[(first + k) % ring_len] ([ring_len][N,A,2]) exactly as pgd_step would -- auto-reset
 * included -- and writes its reward / done / flags into slice k of d_reward / d_done / d_flags ([n_steps][N,A]).  The
 * observation (d_obs, may be NULL) is evaluated ONCE, for the state after the last step: the intermediate steps run without
 * the observation part of the kernel.  n_steps launches on the stream, no host synchronisation in between; the closed loop
 * policy -> action -> step needs pgd_step (the reference's env.step has no counterpart of this call; its decision_repeat is
 * the five physics sub-steps inside every step). */
int pgd_step_n(pgd_handle h, const float* d_action_ring, int ring_len, int first, int n_steps, float* d_obs /*[N,A,D]*/,
               float* d_reward /*[n_steps][N,A]*/, uint8_t* d_done /*[n_steps][N,A]*/, uint32_t* d_flags /*[n_steps][N,A]*/);

/* The same step with the env's results written as ONE packed fp32 row per env, the unit of the per-step gather that
 * BASELINE.json's north star names (envs shard across GPUs, one gather of (obs, reward, done) per step):
 *   d_rows[e * row_stride + ...] = [A*D observation floats | A rewards | A done flags as 0.0 / 1.0], row_stride >= A*(D+2).
 * The kernel writes the row itself -- d_rows is typically this rank's slice of the gather's receive buffer, so no copy
 * kernel packs anything.  d_reward / d_done / d_flags are written as by pgd_step (rank-local bookkeeping). */
int pgd_step_packed(pgd_handle h, const float* d_actions, float* d_rows /*[N,row_stride]*/, int row_stride,
                    float* d_reward /*[N,A]*/, uint8_t* d_done /*[N,A]*/, uint32_t* d_flags /*[N,A]*/);

/* Asynchronous env groups (double-buffered sampling: the policy of group A runs while group B steps).  The reference runs
 * one env per process and lets the RL library interleave processes; here one handle is split into n_groups equal, contiguous
 * groups of envs, each with an internal stream.  pgd_step_group steps ONLY the envs of `group` on that group's stream --
 * consecutive steps of different groups overlap on the GPU (a launch ends with its slowest wave; another group's step fills
 * that tail).  All pointers address the FULL [N, ...] arrays; only the group's rows are read / written.  pgd_group_stream
 * hands out the stream so that the caller can order its own kernels (the policy) with the group's steps; work submitted
 * through pgd_step / pgd_reset (engine stream) is NOT ordered against the group streams: synchronise when switching.
 * An engine in throughput mode (>= 32768 envs: three whole envs per wave) whose group size is not a whole number of such waves is
 * switched back to one env per wave FOR THE REST OF ITS LIFE (same results, 6 - 18 % slower at that size); the switch happens only
 * when the call succeeds (PGD_ERR_ARG leaves the engine unchanged) and pgd_describe_step reports it from then on. */
int pgd_set_groups(pgd_handle h, int n_groups);   /* N % n_groups == 0; 1 = back to a single group */
int pgd_step_group(pgd_handle h, int group, const float* d_actions /*[N,A,2]*/, float* d_obs /*[N,A,D]*/, float* d_reward,
                   uint8_t* d_done, uint32_t* d_flags);
int pgd_group_stream(pgd_handle h, int group, void** hip_stream);
int pgd_group_sync(pgd_handle h, int group);

/* Scripted lane-keeping policy for the controlled agent of a single-agent engine -- what the reference's examples use their
 * trained PPO expert for (examples/ppo_expert, tests/test_functionality/test_expert_performance.py): an action stream that
 * keeps the ego DRIVING, for benchmarks and soak runs.  One launch on the engine stream; reads the rows pgd_step wrote
 * (default state layout, side_lasers == 0: [0] [1] distances to the left / right road edge, [2] heading alignment, [3] speed):
 *   steering = clip(k_lat * 18 * (o[0] - o[1]) / 10 + k_head * (2 o[2] - 1) + noise * n1, -1, 1)
 *   throttle = clip(0.3 * (v_target_kmh - v_kmh) + noise * n2, -1, 1)      n1, n2 ~ U(-1, 1) from the counter RNG (seed, env, tick)
 * d_actions [N, 1, 2] is then handed to pgd_step.  Not part of the reference's env.step: a convenience of this library. */
/* The same policy evaluated by the step itself: pgd_step_lane_keep = pgd_lane_keep_actions on the rows in d_obs (what the previous
 * pgd_step / pgd_reset / pgd_step_lane_keep wrote there) followed by pgd_step with those actions, d_obs rewritten -- as ONE launch on
 * engines with one env per wave (the step kernel reads the four floats where it would read the caller's action; the same arithmetic,
 * the same bits), as the two launches otherwise.  Closed-loop driving without a second launch per step (the policy kernel alone
 * was a fifth of an iteration: its launch floor). */
int pgd_step_lane_keep(pgd_handle h, float k_lat, float k_head, float v_target_kmh, float noise, uint32_t tick, float* d_obs /*[N,1,D]*/,
                       float* d_reward, uint8_t* d_done, uint32_t* d_flags);
int pgd_lane_keep_actions(pgd_handle h, const float* d_obs /*[N,1,D]*/, float* d_actions /*[N,1,2]*/, float k_lat, float k_head,
                          float v_target_kmh, float noise, uint32_t tick);

/* Which step kernel the last pgd_step / pgd_step_packed / pgd_step_group call of this engine launched, as text ("" before the
 * first step): one env per wave, several envs per wave, throughput mode, two waves per env, and whether it is the instantiation
 * specialised for the reference's default single-agent configuration.  For benchmark lines and tests; no reference counterpart. */
int pgd_describe_step(pgd_handle h, char* buf, int cap);

/* Checkpoint / resume (BaseVehicle.get_state/set_state, base_vehicle.py:683-698): raw SoA state blobs.
 * Layout: nf float fields then ni int fields, each [N*V]; query sizes with pgd_state_dims. HOST buffers. */
int pgd_state_dims(pgd_handle h, int* n_float_fields, int* n_int_fields, int* n_env_int_fields);
int pgd_get_state(pgd_handle h, float* h_f, int32_t* h_i, int32_t* h_env_i);
int pgd_set_state(pgd_handle h, const float* h_f, const int32_t* h_i, const int32_t* h_env_i);
/* Recompute localisation + observation from the current state without stepping (engine.after_step + observe,
 * base_env.py:295-301). */
int pgd_observe(pgd_handle h, float* d_obs);

/* Timing helper: HIP-event time [ms] of the last pgd_step on the engine stream.  Off by default (two event packets per
 * step leave idle gaps between back-to-back launches); switch it on with pgd_enable_step_timing(h, 1). */
int pgd_enable_step_timing(pgd_handle h, int on);
int pgd_last_step_ms(pgd_handle h, float* ms);

/* Per-kernel HIP-event profile: between begin and end every pgd_step records events around its two kernels on the
 * engine stream (up to `capacity` steps); end synchronises and returns the average duration of each kernel. */
int pgd_profile_begin(pgd_handle h, int capacity);
/* Strided variant for bench.py.  When pgd_step is a single kernel (observation fused) the two events bracket GROUPS of
 * `stride` consecutive launches and pgd_profile_end reports group time / stride, i.e. the average launch duration with
 * the launches left back to back; `count` is then the number of complete groups.  Otherwise every stride-th step is
 * bracketed on its own. */
int pgd_profile_begin_strided(pgd_handle h, int capacity, int stride);
int pgd_profile_end(pgd_handle h, float* k_step_ms, float* k_observe_ms, int* count);

/* Move the engine to another HIP stream (e.g. the caller's current framework stream, so that a step is ordered like any
 * other op of that stream and needs no events).  Work already enqueued on the previous stream is ordered before anything
 * enqueued on the new one.  The engine never owns `hip_stream`; NULL is the device's default stream. */
int pgd_set_stream(pgd_handle h, void* hip_stream);
int pgd_sync(pgd_handle h);
int pgd_destroy(pgd_handle h);
const char* pgd_version(void);
/* The same network with SPLIT bf16 operands on the bf16 matrix cores (every f32 value = hi + lo, two bf16; a product = three matrix
 * instructions; f32 accumulation): 1/5 of the exact form's matrix time, ~3e-5 of error on an action in [-1, 1] instead of ~1e-6.  The
 * weights are prepared ONCE per policy update into a device buffer of pgd_mlp_prepared_bytes(in_dim) bytes (16-byte aligned):
 * pgd_mlp_prepare splits them and lays them out in the kernel's fragment order (asynchronous on the engine's stream);
 * pgd_mlp_policy_prepared then evaluates the network like pgd_mlp_policy (same rows, groups, action layout). */
size_t pgd_mlp_prepared_bytes(int in_dim);
int pgd_mlp_prepare(pgd_handle h, int in_dim, int hidden, const float* d_w1, const float* d_b1, const float* d_w2, const float* d_b2,
                    const float* d_w3, const float* d_b3, int out_cols, void* d_prepared);
int pgd_mlp_policy_prepared(pgd_handle h, int group, const float* d_obs, int obs_stride, int in_dim, const void* d_prepared, int final_tanh,
                            float* d_actions);
/* Run-time specialisation of the step kernel (no reference counterpart).  The library ships instantiations of k_step with the
 * configurations of the reference's env classes compiled in; any OTHER configuration runs the general kernel, 12 - 17 % behind.
 * pgdrive_amd/jit.py builds, with the same hipcc and flags as the library, a code object that holds k_step with THIS handle's
 * configuration and geometry as literals (every pgd_config field; pgd_step_geometry reports the derived values:
 * N, A, T, V, D, NV, epw, sub, pack_obs, sstride, use_imask, flags: bit 0 objects among the bodies, bit 1 default row layout, bit 2
 * the engine can take such a kernel -- single-agent, one env per wave, scenarios uploaded), and pgd_set_step_module loads it:
 * pgd_step then launches it wherever it would have launched a general kernel, while the engine's geometry and object flag are what
 * the module was built for (pgd_set_groups, a scenario upload that adds objects: back to the library's kernels).  Null path: unload.
 * Results are those of the general kernel to rounding (constant folding re-orders a few fp32 operations), flags and integer state
 * bit-identical: tests/test_parity_gpu.py::test_run_time_kernel_matches_the_general_kernel. */
int pgd_step_geometry(pgd_handle h, int32_t* out12);
int pgd_set_step_module(pgd_handle h, const char* code_object_path, int built_with_objects, int built_with_std_rows);
/* The policy network of a closed loop in one launch: actions[r][0..1] = MLP(obs row r) for every (env, agent) row of the engine
 * (group < 0: all rows, on the engine's stream; group >= 0: the rows of that env group on the group's stream, the twin of
 * pgd_step_group).  Replaces pgdrive/examples/ppo_expert/numpy_expert.py:25-44 (`expert(obs)`: x = tanh(obs @ fc_1 + b);
 * x = tanh(x @ fc_2 + b); out = x @ fc_out + b, the action = the first two outputs) evaluated row by row in numpy, and the
 * policy(obs) call of any rollout loop whose policy is such a network.  Weights: fp32 device arrays, row-major [in][out] as the
 * reference's `kernel` arrays (w1 [in_dim][hidden], w2 [hidden][hidden], w3 [hidden][out_cols]; only columns 0 and 1 of w3 / b3 are
 * used); hidden must be 256; w1, w2, b1, b2 16-byte aligned.  d_obs: rows of obs_stride floats, the first in_dim are the network's input (normally the buffer and
 * row width pgd_step writes).  final_tanh != 0 squashes the two outputs (numpy_expert.py does not; the env clips).
 * d_actions: [rows][2] floats, the layout pgd_step reads.  Asynchronous; may be captured in a HIP graph with the step.
 * Arithmetic: fp32 throughout (the 256-wide layers on the f32 matrix cores: a k-ordered fma chain). */
int pgd_mlp_policy(pgd_handle h, int group, const float* d_obs, int obs_stride, int in_dim, int hidden, const float* d_w1,
                   const float* d_b1, const float* d_w2, const float* d_b2, const float* d_w3, const float* d_b3, int out_cols,
                   int final_tanh, float* d_actions);
/* Multi-agent engines remember, per env, which rows of the LAST observation buffer they were given already hold the zeros of a seat
 * that is not due (identified by the buffer's address and row stride), and do not write them again.  A caller that hands pgd_step
 * a buffer whose address a FORMER buffer had (a caching allocator re-using a freed block: torch.empty per step) calls this first:
 * every row that is not due is then written once more.  Asynchronous on the engine's stream.  pgdrive_amd.Engine.step(out=...) calls
 * it whenever `out` is a tensor it has not seen alive.  (No reference counterpart: the reference returns fresh numpy arrays,
 * base_env.py:303-344.) */
int pgd_forget_rows(pgd_handle h);
/* Identity of the binary: sha256 (16 hex digits) over the sources it was compiled from, written in by pgdrive_amd/build.py
 * ("unstamped" for any other build).  Profile summaries under profiles/ carry the same stamp; bench.py quotes a counter pass only
 * when the stamps agree.  (No reference counterpart: the reference ships no native binary on this path.) */
const char* pgd_source_sha(void);

/* Top-down (bird's-eye) multi-channel observation: TopDownMultiChannel.observe (obs/top_down_obs_multi_channel.py:18-280) of
 * TopDownPGDriveEnv (envs/top_down_env.py:28-42) as a rasteriser kernel.  Image [N, R, R, 2 + frame_stack] float32 in [0, 1]:
 * channel 0 road network (lane lines, route lanes), 1 past ego positions, 2.. the other vehicles now and frame_skip, 2 *
 * frame_skip ... steps ago, the ego at the centre heading up, +-distance metres.  pgd_observe_topdown is called ONCE after
 * every pgd_step (and after pgd_reset): it appends the present state to the per-env history it draws the older frames from.
 * Single-agent engines only, like the reference.  Exact definition: pgdrive_amd/csrc/pgd_topdown.h. */
typedef struct pgd_topdown_config {
  int32_t resolution;       /* R: 84 (top_down_env.py:19) */
  float distance;           /* 30 m */
  int32_t frame_stack;      /* 3 traffic frames */
  int32_t post_stack;       /* 5 past positions */
  int32_t frame_skip;       /* 5 steps between stacked entries */
  int32_t mode;             /* 0: TopDownMultiChannel [R, R, 2 + frame_stack] (TopDownPGDriveEnv / V2, top_down_env.py:28-60);
                               1: TopDownObservation, one RGB frame [R, R, 3] / 255 -- lane lines (35, 35, 35), the ego GREEN
                               (50, 200, 0), the other vehicles BLUE (100, 200, 255) (TopDownSingleFramePGDriveEnv,
                               top_down_env.py:8-26, obs/top_down_obs.py:22-240; the stack / skip fields are not read) */
} pgd_topdown_config;
int pgd_topdown_channels(const pgd_topdown_config* cfg);
int pgd_topdown_enable(pgd_handle h, const pgd_topdown_config* cfg);
int pgd_observe_topdown(pgd_handle h, float* d_img /*[N,R,R,C]*/);
/* The same image as bytes in [0, 255]: the reference's `rgb_clip=False` (pgdrive_env.py:133-141; top_down_obs_multi_channel.py:208-211,
 * 253-256, 277-280 return the uint8 pygame values instead of float32 / 255).  byte = (int)(float value x 255): (line texels x 35 + route
 * texels x 64) / 2 of the pixel's 2 x 2 cell on the road channel, 255 at a past position, 176 inside a vehicle box; RGB frame: lines 35,
 * the ego (50, 200, 0), the others (100, 200, 255).  A quarter of the bytes of the float image (a write-bound kernel: DESIGN.md
 * section 14).  `d_img` 16-byte aligned.  One call advances the pose history like pgd_observe_topdown: call ONE of the two per step. */
int pgd_observe_topdown_u8(pgd_handle h, uint8_t* d_img /*[N,R,R,C]*/);

/* ---------------------------------------------------------------------------------------------------------------------
 * Per-step gather by direct peer writes (multi-GPU, one process per GPU).  The reference has no distributed layer (one env
 * per process, engine_utils.py:8-15); BASELINE.json's north star shards the envs over the GPUs of a node with one gather of
 * (obs, reward, done) per step.  xGMI is point to point, so instead of a ring every rank writes its packed rows (the rows
 * pgd_step_packed produces) into the receive buffers of all peers at once.  Each rank owns `nbuf` receive buffers of
 * [world * n_rows][row_floats] fp32 (rank r's rows at row offset r * n_rows) plus a small flag area, in one device block
 * that peers map over HIP IPC.  Sequence numbers start at 1 and grow by 1 per push; buffer use is round robin.
 *   create  -> export (handle blob, exchanged by the host, e.g. torch.distributed.all_gather_object) -> connect per peer
 *   per step: pgd_step_packed(d_rows = own slice of pgd_gather_buffer(buf)) ; pgd_gather_push(buf, seq)
 *   consumer: pgd_gather_wait(buf, seq) ... read the buffer ... pgd_gather_release(buf, seq)
 * All three are asynchronous on the given stream; a peer that never arrives sets the status word instead of hanging.
 * seq == 0 takes the sequence number from a per-buffer counter on the device (advanced by pgd_gather_release): the calls of a step
 * are then identical every time, and  wait, release, pgd_step_packed, push  of a multiple of nbuf consecutive steps can be captured
 * in one HIP graph (a handle uses either host-counted or device-side sequences, not both). */
#define PGD_GATHER_HANDLE_BYTES 64
typedef struct pgd_gather* pgd_gather_handle;
int pgd_gather_create(int device, int world, int rank, int n_rows, int row_floats, int nbuf, pgd_gather_handle* out);
int pgd_gather_buffer(pgd_gather_handle g, int buf, float** d_recv /* [world*n_rows, row_floats] */);
int pgd_gather_export(pgd_gather_handle g, void* h_handle /* PGD_GATHER_HANDLE_BYTES */);
int pgd_gather_connect(pgd_gather_handle g, int peer, const void* h_handle);
int pgd_gather_push(pgd_gather_handle g, int buf, int seq, void* hip_stream);
int pgd_gather_wait(pgd_gather_handle g, int buf, int seq, void* hip_stream);
int pgd_gather_release(pgd_gather_handle g, int buf, int seq, void* hip_stream);
int pgd_gather_status(pgd_gather_handle g, int* err /* 0 = ok, 1 = an ack never came, 2 = rows never came */);
/* 1: the receive block is fine-grained (device-coherent across agents) memory, as the protocol wants; 0: the runtime refused
 * hipExtMallocWithFlags(..., hipDeviceMallocFinegrained) or PGD_GATHER_COARSE was set and the block is plain hipMalloc memory. */
int pgd_gather_mem_kind(pgd_gather_handle g, int* fine_grained);
int pgd_gather_destroy(pgd_gather_handle g);

#ifdef __cplusplus
}
#endif
#endif
