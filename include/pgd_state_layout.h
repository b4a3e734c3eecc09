/*
 * pgd_state_layout.h — field order of the struct-of-arrays simulation state exchanged through
 * pgd_get_state / pgd_set_state (checkpoint / resume; BaseVehicle.get_state/set_state, base_vehicle.py:683-698).
 *
 * Float fields are [PGD_NF][N*V], int fields [PGD_NI][N*V] (index env*V + slot), env ints [PGD_NEI][N].
 */
#ifndef PGD_STATE_LAYOUT_H
#define PGD_STATE_LAYOUT_H

enum {
  SF_X = 0, SF_Y, SF_THETA,   /* position, heading_theta [rad]          base_vehicle.py:390-416 */
  SF_SPEED,                   /* [m/s] (speed property is km/h)         base_vehicle.py:394-401 */
  SF_STEER, SF_THROTTLE,      /* last applied action                    base_vehicle.py:343-349 */
  SF_LASTX, SF_LASTY,         /* last_position                          base_vehicle.py:246 */
  SF_LASTHX, SF_LASTHY,       /* last_heading_dir                       base_vehicle.py:247 */
  SF_ACT0S, SF_ACT0T,         /* last_current_action[0] (older)         base_vehicle.py:171,248 */
  SF_ACT1S, SF_ACT1T,         /* last_current_action[1] (newer) */
  SF_PID_HP, SF_PID_HI,       /* IDM heading PID p_error, i_error       PID_controller.py:1-17 */
  SF_PID_LP, SF_PID_LI,       /* IDM lateral PID.  Controlled agents have no PID; under PGD_MA_TOLLGATE the four fields hold
                                 TollGateObservation.in_toll_time and StayTimeManager's entry step, exit step (-1 = none) and
                                 last block id char (-1 = not seen yet)  marl_tollgate.py:36-60,76-96; under PGD_MA_PARKING
                                 SF_PID_HP is 1 + the agent's destination parking space (v_dest_pair), 0 = none */
  SF_TARGET_SPEED,            /* IDMPolicy.target_speed [km/h]          idm_policy.py:182 */
  SF_ENERGY,                  /* energy_consumption                     base_vehicle.py:278-290 */
  SF_DIST_LEFT, SF_DIST_RIGHT,/* dist_to_left_side / right_side         base_vehicle.py:380-388 */
  SF_EP_REWARD,               /* episode_rewards                        base_env.py:335-339 */
  SF_AGENT_ID,                /* multi-agent: k of "agent{k}" (integer-valued)  agent_manager.py:154-175 */
  SF_HX, SF_HY,               /* unit heading vector (cos, sin of SF_THETA) as the engine carries it from step to step (the
                                 physics advances it by rotations instead of calling sincos per step).  Part of the checkpoint
                                 so that get_state -> set_state resumes bit-exactly; pgd_set_state keeps it only while it agrees
                                 with SF_THETA to 1e-4 (an edited THETA, or a state built by hand, gets cos / sin of THETA).  SF_THROTTLE is returned equal to SF_ACT1T: the last
                                 applied throttle IS the newer entry of the action deque (base_vehicle.py:343-349) */
  PGD_NF
};
enum {
  SI_STATUS = 0,              /* ST_* */
  SI_LANE,                    /* vehicle.lane (map-local id) */
  SI_CK0, SI_CK1,             /* Navigation._target_checkpoints_index   navigation.py:132 */
  SI_RLANE,                   /* traffic: IDMPolicy.routing_target_lane (-1 = None); agents: episode_length (base_env.py:338) */
  SI_TIMER,                   /* traffic: IDMPolicy.overtake_timer; dying agents: delay-done countdown (agent_manager.py:191-199) */
  SI_VFLAGS,                  /* PGD_F_* vehicle state bits */
  SI_SPAWN,                   /* index of the slot's pgd_spawn record inside its scenario (slot id, or a respawn record) */
  PGD_NI
};
enum {
  EI_SCEN = 0,                /* scenario id of the running episode */
  EI_NEXT_GROUP,              /* next traffic trigger group             traffic_manager.py:78-85 */
  EI_EP_STEPS,                /* episode_steps                          base_env.py:185 */
  EI_EPISODES,                /* auto-reset count (RNG counter) */
  EI_STEPS_TOTAL,             /* steps since pgd_reset (RNG counter) */
  EI_NEXT_AGENT,              /* next "agent{k}" id (AgentManager.next_agent_count) */
  EI_AUX,                     /* PGD_MA_PARKING: ParkingLotSpawnManager.parking_space_available as a bit mask */
  EI_NEAR,                    /* device-private hints (pgd_get_state returns 0, pgd_set_state ignores it and resets it to "unknown"):
                                 bit 0, left by the fused observation: 0 = no body can reach an agent during the next step (contact
                                 tests skipped); bits 1-2, single-agent one-env waves: 1 / 2 = the agent does not / does stand on the
                                 trigger road of the next traffic group (the next step's trigger test), 0 = unknown */
  PGD_NEI
};
enum { ST_EMPTY = 0, ST_PENDING = 1, ST_ACTIVE = 2, ST_REMOVED = 3, ST_DYING = 4 /* finished agent, static, counting down */ };

#endif
