// pgd_dynamics.h -- bicycle dynamics, vehicle reset, reward / done.
// Part of the single translation unit pgd_engine.hip (included there, in this order, after pgd_device.h).
#ifndef PGD_DYNAMICS_H
#define PGD_DYNAMICS_H

// Fused multiply-adds are WRITTEN OUT in this file's arithmetic (`fmaf`) and the compiler's own contraction / re-association is
// switched off inside these routines: the instantiations of k_step (one env per wave, several envs per wave, specialised, general) inline
// the same source, and left to itself the compiler fuses `a * b + c * d` one way here and the other way there -- a heading vector
// that differs in the last bit after one step, poses that drift apart by ulps per step, and ray columns of the two modes that
// differed by up to 1.4e-5 after ninety steps (tools/mode_diff.py, profiles/r06_notes.md).  Same instruction
// count as the fused forms the compiler chose; every instantiation now produces the same bits from the same state.

// tan on |x| <= 1 rad (steering locks are 35-50 deg; pgd_upload_scenarios rejects max_steer > 1) as sin / cos from their
// Taylor polynomials (truncation < 2e-10): the error is the fp32 rounding of the quotient, at a sixth of tanf's instructions
DEV float tan_small(float x) {
#pragma clang fp contract(off)
#pragma clang fp reassociate(off)
  const float q = x * x;
  const float sn = x * fmaf(q, fmaf(q, fmaf(q, fmaf(q, fmaf(q, -1.0f / 39916800.0f, 1.0f / 362880.0f), -1.0f / 5040.0f), 1.0f / 120.0f), -1.0f / 6.0f), 1.0f);
  const float cs = fmaf(q, fmaf(q, fmaf(q, fmaf(q, fmaf(q, fmaf(q, 1.0f / 479001600.0f, -1.0f / 3628800.0f), 1.0f / 40320.0f), -1.0f / 720.0f), 1.0f / 24.0f), -0.5f), 1.0f);
  return sn / cs;
}

// kinematic bicycle (component/highway_vehicle/kinematics.py:134-156) driven by the reference's action -> force mapping
// (base_vehicle.py:343-376); see DESIGN.md §3 for the substitution of Bullet's raycast vehicle.
// keep (wave-uniform where one wave carries one env): record the pose after every sub-step and the path length for the
// contact test of this step.  ONE compiled body for both cases: two instantiations would contract their multiply-adds
// differently and an env would not step bit-identically with and without the bookkeeping.
DEV void dynamics(const PgdDev& d, const pgd_spawn& p, Veh& r, bool reverse, const float thr, struct SubPose* sub, int slot, const bool keep) {
#pragma clang fp contract(off)
#pragma clang fp reassociate(off)
  float dt = d.cfg.dt;
  float force = 0.0f, brake = 0.0f;
  if (thr >= 0.0f) {
    brake = 2.0f;
    force = (fabsf(r.v) * 3.6f > p.max_speed) ? 0.0f : p.max_engine_force * thr;
  } else if (reverse) {  // enable_reverse: engine force backwards, no brake (base_vehicle.py:370-373)
    force = p.max_engine_force * thr;
  } else {
    brake = fabsf(thr) * p.max_brake_force;
  }
  float delta = -clipf(r.steer, -1.0f, 1.0f) * p.max_steer;
  // beta = atan(t), t = tan(delta)/2  ->  cos(beta) = 1/sqrt(1+t^2), sin(beta) = t/sqrt(1+t^2)
  float t = 0.5f * tan_small(delta);
  float cb = 1.0f / sqrtf(fmaf(t, t, 1.0f)), sb = t * cb;
  // unit vector of the motion direction th + beta, advanced by exact small-angle rotations instead of sincos per sub-step
  float cd = fmaf(r.hx, cb, -(r.hy * sb)), sd = fmaf(r.hy, cb, r.hx * sb);
  float inv_half_base = 2.0f / p.wheelbase;
  float dv_brake = fminf((4.0f * brake) / p.mass, (p.friction * 9.81f) * dt);
  float dv_engine = ((4.0f * force) / p.mass) * dt;
  float trav = 0.0f;
  const int n_mid = (keep && d.cfg.decision_repeat <= PGD_MAX_SUB) ? d.cfg.decision_repeat - 1 : 0;  // sub-step poses kept
  for (int k = 0; k < d.cfg.decision_repeat; ++k) {
    trav = fmaf(fabsf(r.v), dt, trav);
    const float vd = r.v * dt;  // distance of the sub-step
    r.x = fmaf(vd, cd, r.x);
    r.y = fmaf(vd, sd, r.y);
    float dth = (vd * sb) * inv_half_base;  // |dth| < 0.25 rad at 80 km/h and full lock
    r.th += dth;
    float q = dth * dth;
    float sn = dth * fmaf(q, fmaf(q, fmaf(q, -1.0f / 5040.0f, 1.0f / 120.0f), -1.0f / 6.0f), 1.0f);
    float cs = fmaf(q, fmaf(q, fmaf(q, fmaf(q, 1.0f / 40320.0f, -1.0f / 720.0f), 1.0f / 24.0f), -0.5f), 1.0f);
    float ncd = fmaf(cd, cs, -(sd * sn));
    sd = fmaf(sd, cs, cd * sn);
    cd = ncd;
    if (force != 0.0f) r.v += dv_engine;
    else r.v = r.v >= 0.0f ? fmaxf(0.0f, r.v - dv_brake) : fminf(0.0f, r.v + dv_brake);
    if (!reverse) r.v = fmaxf(r.v, 0.0f);
    if (k < n_mid) {  // pose after this sub-step
      if (sub) sub->p[k][slot] = make_float4(r.x, r.y, cd, sd);
    }
  }
  if (keep) {
    if (sub) { sub->trav[slot] = trav; sub->beta[slot] = make_float2(cb, sb); }
  }
  // heading unit vector = motion direction rotated back by beta, renormalised
  float hx = fmaf(cd, cb, sd * sb), hy = fmaf(sd, cb, -(cd * sb));
  float inv = 1.0f / sqrtf(fmaf(hx, hx, hy * hy));
  r.hx = hx * inv;
  r.hy = hy * inv;
  // heading_theta as the reference reports it; keeps the fp32 angle exact to an ulp of pi however often the car has turned
  // (the carried heading vector above is what the geometry uses)
  r.th = heading_wrap(r.th);
}

DEV void reset_vehicle(const pgd_spawn& p, const float2 hv, Veh& r, int spawn_index, bool is_agent) {  // base_vehicle.py:292-339
  memset(&r, 0, sizeof(Veh));
  // agents have no PID state: under PGD_MA_TOLLGATE the fields carry in_toll_time = 0 and entry / exit / last block = none
  // (marl_tollgate.py:36-60,76-96); harmless otherwise
  r.spawn = spawn_index;
  r.rlane = is_agent ? 0 : -1;  // agents: episode length; traffic: IDMPolicy.routing_target_lane = None
  r.hx = 1.0f;
  if (p.lane < 0) { r.status = ST_EMPTY; return; }
  if (is_agent) { r.php = (float)p.aux; r.phi = -1.0f; r.plp = -1.0f; r.pli = -1.0f; }  // php: parking destination / toll time
  r.status = p.group == -1 ? ST_ACTIVE : ST_PENDING;  // PGD_GROUP_NEVER (-2): in the world, never driven
  r.x = p.x; r.y = p.y; r.th = heading_wrap(p.heading);
  r.lastx = p.x; r.lasty = p.y;
  r.lasthx = hv.x; r.lasthy = hv.y;
  r.hx = r.lasthx; r.hy = r.lasthy;
  r.target = 30.0f;
  r.lane = p.lane;
  r.ck0 = 0;
  r.ck1 = p.n_ckpt > 2 ? 1 : 0;
  r.timer = p.timer0;
}

// reward / done: envs/pgdrive_env.py:162-258, base_vehicle.py:738-745
// MARL = false: the single-agent env; none of the multi-agent reward / out-of-road variants (marl_flags == 0) is compiled in.
template <bool MARL, class MV>
// `fl` = the record of the destination lane sp.dest_lane (the caller may have read it ahead of time)
DEV float reward_done(const PgdDev& d, const MV& mv, const pgd_spawn& sp, const pgd_lane& fl, const Veh& r, const RouteCtx& ctx,
                      unsigned& flags_out, bool& done_out) {
  const pgd_config& g = d.cfg;
  const int mflags = MARL ? g.marl_flags : 0;
  unsigned vf = (unsigned)r.vflags;
  const float positive = ctx.positive;
  float w = mv.lane_width();
  float reward = ctx.drive;  // formed by after_step_vehicle from the coordinates it had just evaluated
  if (mflags & PGD_MA_TOLLGATE) {  // MultiAgentTollgateEnv.reward_function (marl_tollgate.py:195-232)
    if (r.blk == '$') {
      // BaseVehicle.overspeed (base_vehicle.py:759-761): lane.speed_limit (3 on toll lanes, 1000 elsewhere) < speed [km/h]
      const bool lane_toll = mv.roads()[mv.lanes[r.lane].road].block_id == '$';
      if (lane_toll && 3.0f < speed_kmh(r.v)) reward = -g.overspeed_penalty * speed_kmh(r.v) / sp.max_speed;
    } else reward += g.speed_reward * (speed_kmh(r.v) / sp.max_speed);
  } else
  reward += g.speed_reward * (speed_kmh(r.v) / sp.max_speed) * positive;
  unsigned out = vf & (PGD_F_ON_YELLOW | PGD_F_ON_WHITE | PGD_F_ON_BROKEN | PGD_F_CRASH_SIDEWALK | PGD_F_OFF_LANE |
                       PGD_F_OUT_OF_ROUTE | PGD_F_CRASH_VEHICLE | PGD_F_CRASH_OBJECT | PGD_F_CRASH_BUILDING);
  float lon, lat;
  lane_local(fl, r.x, r.y, lon, lat);
  bool arrive = (fl.length - 5.0f < lon && lon < fl.length + 5.0f) && (w * 0.5f >= lat && lat >= (0.5f - (float)r.cur_n) * w);
  unsigned oor_bits = (mflags & PGD_MA_TOLLGATE) ? PGD_F_CRASH_SIDEWALK  // marl_tollgate.py:234-240
                      : (mflags & PGD_MA_PARKING) ? (PGD_F_OFF_LANE | PGD_F_CRASH_SIDEWALK)  // marl_parking_lot.py:213-217
                                                        : (PGD_F_ON_WHITE | PGD_F_OFF_LANE | PGD_F_CRASH_SIDEWALK);
  if (!(mflags & PGD_MA_YELLOW_OK)) oor_bits |= PGD_F_ON_YELLOW;
  bool oor = (vf & oor_bits) != 0;
  if (g.out_of_route_done) oor = oor || (vf & PGD_F_OUT_OF_ROUTE);
  bool crash = (vf & PGD_F_CRASH_VEHICLE) != 0, crash_obj = (vf & PGD_F_CRASH_OBJECT) != 0;
  if (arrive) out |= PGD_F_ARRIVE;
  if (oor) out |= PGD_F_OUT_OF_ROAD;
  if (arrive) reward = g.success_reward;
  else if (oor) reward = -g.out_of_road_penalty;
  else if (crash) reward = -g.crash_vehicle_penalty;
  else if (crash_obj) reward = -g.crash_object_penalty;
  flags_out = out;
  done_out = arrive || oor || crash || crash_obj || (vf & PGD_F_CRASH_BUILDING) != 0;  // pgdrive_env.py:162-194
  // SafePGDriveEnv.done_function (safe_pgdrive_env.py:49-56): a step with crash_vehicle, else crash_object, is not terminal
  if (g.safe_rl_env && (crash || crash_obj)) done_out = false;
  return reward;
}

#endif
