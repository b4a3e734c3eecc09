// pgd_observe.h -- observation: state + navigation block, detector fans, neighbour rows, lidar.
// Part of the single translation unit pgd_engine.hip (included there, in this order, after pgd_device.h).
#ifndef PGD_OBSERVE_H
#define PGD_OBSERVE_H
#ifdef PGD_NT_OBS
#define OBS_ST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define OBS_ST(p, v) (*(p) = (v))
#endif
// the fans of the multi-agent rows (written once per step from LDS, read by nobody on this chip's L2 again): streaming stores
#ifdef PGD_PLAIN_FAN
#define OBS_ST_FAN(p, v) (*(p) = (v))
#else
#define OBS_ST_FAN(p, v) __builtin_nontemporal_store((v), (p))
#endif
typedef float obs_f2 __attribute__((ext_vector_type(2)));
typedef float obs_f4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------------
// observation: LidarStateObservation.observe (obs/state_obs.py:132-170) for one (env, agent)
// ---------------------------------------------------------------------------------------------------------------------
DEV void navi_info_for(const LaneNav& ref, float w, int n_cur, float px, float py, float hx, float hy, float* out) {
  // Navigation._get_info_for_checkpoint (navigation.py:213-260); ref = ref_lanes[0] of the checkpoint's road
  float later_middle = ((float)n_cur * 0.5f - 0.5f) * w;
  float dx = ref.ex + later_middle * ref.nx - px, dy = ref.ey + later_middle * ref.ny - py;
  float dn = norm2(dx, dy);
  if (dn > 50.0f) { dx = dx / dn * 50.0f; dy = dy / dn * 50.0f; }
  float ph, ps;
  projection(hx, hy, dx, dy, ph, ps);
  float bend = 0.0f, dir = 0.0f, angle = 0.0f;
  if (ref.dir != 0.0f) {
    bend = ref.radius / (60.0f + n_cur * w);
    dir = ref.dir;
    angle = ref.angle;
  }
  out[0] = clipf((ph / 50.0f + 1.0f) * 0.5f, 0.0f, 1.0f);
  out[1] = clipf((ps / 50.0f + 1.0f) * 0.5f, 0.0f, 1.0f);
  out[2] = clipf(bend, 0.0f, 1.0f);
  out[3] = clipf((dir + 1.0f) * 0.5f, 0.0f, 1.0f);
  out[4] = clipf((angle * (180.0f / PGD_PI) / 135.0f + 1.0f) * 0.5f, 0.0f, 1.0f);
}

DEV float heading_diff(const pgd_lane& l, float px, float py, float fx, float fy) {  // base_vehicle.py:433-458
  float lx, ly;
  if (l.dir == 0.0f) { lx = -l.by; ly = l.bx; }
  else if (l.dir < 0.0f) { lx = px - l.ax; ly = py - l.ay; }
  else { lx = l.ax - px; ly = l.ay - py; }
  float ln = norm2(lx, ly), fn = norm2(fx, fy);
  if (ln * fn == 0.0f) return 0.0f;
  return clipf((fx * lx + fy * ly) / (ln * fn), -1.0f, 1.0f) * 0.5f + 0.5f;
}

#define PGD_LIDAR_MINB_WORDS (6 * WAVE)  // LDS words behind `minb` of observe_agent: per-beam minima, then the rounds' start masks
#ifndef PGD_LIDAR_INC_MIN
#define PGD_LIDAR_INC_MIN 2  // bodies in the lidar broad phase from which a one-wave row casts incidences instead of rounds
#endif
struct ObsLds {  // bodies inside the lidar broad phase of the observing agent, compacted
  float bx[MAXV], by[MAXV], bux[MAXV], buy[MAXV], bhl[MAXV], bhw[MAXV], bspd[MAXV];
  float bdist[MAXV];  // centre distance; +inf for traffic objects, which are never ranked as neighbour vehicles
  int bi0[MAXV], bcnt[MAXV];  // lidar beams [bi0, bi0 + bcnt) mod num_lasers that can reach the body (conservative)
  unsigned bsec[MAXV];        // bit q: that window meets the beams [64 q, 64 q + 63] -- what one wave casts in one round
  int bslot[MAXV];            // slot of the body in its env
  int bpref[MAXV];            // incidence form of the lidar: first incidence of the body's window
  int rank_slot[16];          // PGD_MA_OTHERS_STATE: slot of the neighbour of rank r (-1 = none) ...
  float rank_spd[16];         // ... and its speed [km/h] as the observer sees it (0 for a static finished agent)
  int n, nveh;
};
union ObsScratch {
  ObsLds ol;
  SubPose sp;
};
struct AgentView {  // what the observation needs from the observing vehicle
  float x, y, th, hx, hy, dl, dr, v, steer, a0s, a0t, lhx, lhy;
  int cur_first, cur_n, next_first;  // RouteCtx of the vehicle
  int blk;                           // block id char of its current road
  float toll_time;                   // TollGateObservation.in_toll_time (PGD_MA_TOLLGATE)
  int env, slot;                     // for the lidar noise stream
  uint32_t tick;                     // steps since pgd_reset
};

// one wave compacts the candidates: lane `o` brings vehicle o of the env (present = in the physics world)
// `near_out` (optional): set when body o can reach the observing agent during the NEXT step -- centre distance within the two
// sizes (half length + half width bounds the circumradius) plus the longest paths both can drive in one step of
// t_step = dt * decision_repeat seconds (speed + 10 m/s^2 of acceleration -- three times what the strongest engine gives --
// 5 % slack: near_reach).  k_step skips its contact tests in envs where no body is near any agent.  The test only sees bodies
// inside the lidar broad phase, so the hint is used only while the lidar range covers every possible reach (near_hint_usable).
DEV float near_reach(float speed_ms, float t_step) { return (fabsf(speed_ms) + 10.0f * t_step) * t_step * 1.05f; }
DEV bool near_hint_usable(const pgd_config& c) {
  const float t_step = c.dt * (float)c.decision_repeat;
  // two bodies at 150 km/h (nothing drives faster: the engine force is cut above max_speed <= 80 km/h) + the two largest
  // circumradius bounds (MAX_LENGTH 10 / 2 + MAX_WIDTH 2.5 / 2 each, base_vehicle.py:83-84)
  return c.num_lasers > 0 && c.lidar_dist >= 2.0f * near_reach(150.0f / 3.6f, t_step) + 12.6f;
}
// atan2 for the beam windows only: minimax polynomial of atan on [0, 1] (error < 2e-5 rad = 1e-3 of a beam at 240 beams: see beam_window), a third of the library routine's instructions.  Never used for an observed value.
DEV float atan2_window(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float z = mx > 0.0f ? mn / mx : 0.0f, q = z * z;
  float r = z * (0.99997726f + q * (-0.33262347f + q * (0.19354346f + q * (-0.11643287f + q * (0.05265332f + q * -0.01172120f)))));
  r = ay > ax ? 0.5f * PGD_PI - r : r;
  r = x < 0.0f ? PGD_PI - r : r;
  return y < 0.0f ? -r : r;
}

// The beams that can reach a body: it lies inside the circle of radius `rad` around its centre (rad = 1.02 x circumradius + 1 cm),
// so only beams within asin(rad / dist) of the centre direction can hit it; asin(q) <= q + (pi/2 - 1) q^3 on [0, 1].  The window
// is the INTEGER beams of [ic - hb, ic + hb]: ceil / floor, not floor / ceil.  What has to be covered is the error of the centre
// angle -- atan2_window's 2e-5 rad and the fp32 rounding of (rx, ry), together < 1e-4 rad -- and the 2 % + 1 cm on the radius
// already widen the half angle by >= 6e-4 rad at any distance inside the lidar range; PGD_WINDOW_SLACK beams are added on top.
// (Rounds 1 - 4 carried 1.5 beams of slack AND rounded outwards: 3 - 5 beams per window that no ray of which could hit, half of
// all incidences of a 72-beam fan.)  The culling never removes a hit: the cloud stays bit-identical to the all-pairs test.
#ifndef PGD_WINDOW_SLACK
#define PGD_WINDOW_SLACK 0.05f
#endif
DEV void beam_window(const float rx, const float ry, const float q, const int NL, int& i0, int& cnt) {
  const float inv_unit = (float)NL * (0.5f / PGD_PI);
  const float ic = atan2_window(ry, rx) * inv_unit, hb = (q + 0.5708f * q * q * q) * inv_unit + PGD_WINDOW_SLACK;
  const int lo = (int)ceilf(ic - hb);
  const int hi = max((int)floorf(ic + hb), lo);  // (never empty: the owner lookup of the incidence paths wants distinct starts)
  if (hi - lo + 1 < NL) {
    cnt = hi - lo + 1;
    i0 = lo % NL;
    if (i0 < 0) i0 += NL;
  }
}

template <bool OBJ>
DEV void obs_compact(ObsLds& L, int o, int a, bool present, bool is_vehicle, float x, float y, float ux, float uy, float hl,
                     float hw, float spd, float px, float py, float R, float hx, float hy, int NL, float ag_reach = 0.0f,
                     bool* near_out = nullptr, float t_step = 0.1f) {
  if (!OBJ) is_vehicle = true;
  bool in = present && o != a && shape_point_dist<OBJ>(Obb{x, y, ux, uy, hl, hw}, px, py) <= R;
  if (near_out) *near_out = false;
  unsigned long long m = __ballot(in), mv_ = OBJ ? __ballot(in && is_vehicle) : m;
  if (in) {
    int k = __popcll(m & ((1ull << o) - 1ull));
    L.bx[k] = x; L.by[k] = y; L.bux[k] = ux; L.buy[k] = uy; L.bhl[k] = hl; L.bhw[k] = hw; L.bspd[k] = spd;
    const float dist = norm2(px - x, py - y);
    L.bdist[k] = is_vehicle ? dist : __builtin_inff();
    // (a body that can reach the agent within a step is inside the lidar broad phase a fortiori: R >= 20 m in every config)
    if (near_out) *near_out = dist <= ag_reach + hl + (hw < 0.0f ? 0.0f : hw) + near_reach(spd * (1.0f / 3.6f), t_step) + 0.05f;
    // the body lies inside the circle of radius rad around its centre: the beams that can reach it (beam_window)
    const float rad = (hw < 0.0f ? hl : norm2(hl, hw)) * 1.02f + 0.01f;
    int i0 = 0, cnt = NL;
    if (dist > rad && NL > 0) {
      const float rx = (x - px) * hx + (y - py) * hy, ry = (y - py) * hx - (x - px) * hy;  // centre in the vehicle frame
      beam_window(rx, ry, rad / dist, NL, i0, cnt);
    }
    L.bi0[k] = i0; L.bcnt[k] = cnt;
    {  // sectors of 64 consecutive beams the window reaches (more than 32 sectors: every one)
      unsigned sec = 0xffffffffu;
      if (cnt < NL && NL <= 2048) {
        sec = 0u;
        const int last = i0 + cnt - 1;  // the window is [i0, last], possibly past NL - 1 (wraps)
        for (int q = 0; q * 64 < NL; ++q) {
          const int lo = q * 64, hi = min(lo + 63, NL - 1);
          const bool hit = (i0 <= hi && last >= lo) || (last >= NL && last - NL >= lo);
          sec |= hit ? (1u << q) : 0u;
        }
      }
      L.bsec[k] = sec;
    }
    L.bslot[k] = o;
  }
  if (o == 0) { L.n = __popcll(m); L.nveh = __popcll(mv_); }
}

// What a thread of a row reads from memory at addresses the observing vehicle's view fixes: the lane record (heading_diff) or the
// navigation extract its state float needs, and the beam directions of its first four lidar rounds.  obs_preload issues the reads;
// a caller that has other work to do first (the fused observation of k_step: the compaction of the bodies) calls it before that
// work and hands the result to observe_agent -- read where they are used, each of them is waited for on the spot.
struct ObsPre {
  pgd_lane ml;
  LaneNav nv;
  float2 bd0, bd1, bd2, bd3;
};
DEV int state_lane_of(const AgentView& ag, int q) {  // heading_diff -> last lane of the current road; navi -> first lanes
  return q < 8 ? ag.cur_first + ag.cur_n - 1 : (q < 13 ? ag.cur_first : ag.next_first);
}
template <class MV>
DEV void obs_preload(const PgdDev& d, const MV& mv, const AgentView& ag, int tid, int nt, ObsPre& p) {
  const int NL = d.cfg.num_lasers;
  const int lid = state_lane_of(ag, tid);
  if (tid == 2) p.ml = mv.lanes[lid];
  if (tid >= 8 && tid < 18) p.nv = mv.lnav()[lid];
  p.bd0 = p.bd1 = p.bd2 = p.bd3 = make_float2(0.0f, 0.0f);
  if (NL > 0) {
    p.bd0 = d.beam[min(tid, NL - 1)]; p.bd1 = d.beam[min(nt + tid, NL - 1)];
    p.bd2 = d.beam[min(2 * nt + tid, NL - 1)]; p.bd3 = d.beam[min(3 * nt + tid, NL - 1)];
  }
}

// StateObservation.observe of one vehicle (state_obs.py:42-106): ego state + detector fans + navigation info, written to
// row[0 .. state length) by threads tid in [0, nt).  `pre`: obs_preload's result for this thread (its float q = tid), or null
template <bool STD, class MV>
DEV void state_block(const PgdDev& d, const MV& mv, const pgd_spawn& sp, const AgentView& ag, float* __restrict__ row,
                     int tid, int nt, const ObsPre* pre = nullptr) {
  const float px = ag.x, py = ag.y, hx = ag.hx, hy = ag.hy;
  // StateObservation.vehicle_state (state_obs.py:58-106) + navi info (navigation.py:185-197): one lane per float.
  // Row layout: [side fan k | 2 lateral distances][6 ego floats][lane-line fan m][10 navi][4*NO neighbours][NL beams]
  const int KS = STD ? 0 : d.cfg.side_lasers, KM = STD ? 0 : d.cfg.lane_line_lasers;
  const bool toll = !STD && (d.cfg.marl_flags & PGD_MA_TOLLGATE) != 0;  // no navigation block, 2 toll floats after the lidar
  const int RAM = (!STD && d.cfg.random_agent_model) ? 2 : 0;  // LENGTH / 10, WIDTH / 2.5 after the lane-line fan (state_obs.py:102-105)
  const int o_ego = KS > 0 ? KS : 2, o_navi = o_ego + 6 + KM + RAM;
  if (RAM && tid == nt - 1) {
    row[o_ego + 6 + KM] = clipf(sp.length / 10.0f, 0.0f, 1.0f);
    row[o_ego + 6 + KM + 1] = clipf(sp.width / 2.5f, 0.0f, 1.0f);
  }
  // one lane per float.  Fewer than 18 cooperating threads (the multi-agent observation: 64 / A lanes per agent) would run the
  // whole branch ladder below once per round of nt floats; they take the eight ego floats in the loop and the navigation block as
  // two whole check points (thread 0 and 1, five floats each: one evaluation instead of ten that keep one value each)
  const int n_loop = nt < 18 ? 8 : 18;
  // (few threads: the heading_diff lane of the thread that will take float 2 is read here, together with the check point records
  // of the loop below -- one memory round trip for the block instead of two in a row)
  pgd_lane ml_few;
  const bool few_ml = nt < 18 && !pre && tid == 2 % nt;
  if (few_ml) ml_few = mv.lanes[state_lane_of(ag, 2)];
  if (nt < 18 && !toll)
    for (int which = tid; which < 2; which += nt) {
      const LaneNav nvw = mv.lnav()[state_lane_of(ag, 8 + 5 * which)];
      float out[5];
      navi_info_for(nvw, mv.lane_width(), ag.cur_n, px, py, hx, hy, out);
#pragma unroll
      for (int c = 0; c < 5; ++c) row[o_navi + 5 * which + c] = out[c];
    }
  for (int q = tid; q < n_loop; q += nt) {
    // every lane fetches the one lane record its float needs BEFORE the branch ladder, so the reads overlap instead of
    // queueing behind each other branch by branch: heading_diff -> last lane of the current road; navi -> first lanes
    const int lid = state_lane_of(ag, q);
    pgd_lane ml;  // only the heading_diff lane needs the 64-byte lane record
    LaneNav nv;
    if (pre && q == tid) {
      if (q == 2) ml = pre->ml;
      if (q >= 8) nv = pre->nv;
    } else {
      if (q == 2) ml = few_ml ? ml_few : mv.lanes[lid];
      if (q >= 8) nv = mv.lnav()[lid];
    }
    const float max_speed = sp.max_speed;
    float v = 0.0f;
    int col = -1;
    if (q == 0) { v = clipf(ag.dl / 18.0f, 0.0f, 1.0f); col = KS > 0 ? -1 : 0; }  // (MAX_LANE_NUM+1)*MAX_LANE_WIDTH
    else if (q == 1) { v = clipf(ag.dr / 18.0f, 0.0f, 1.0f); col = KS > 0 ? -1 : 1; }
    else if (q == 2) { v = heading_diff(ml, px, py, hx, hy); col = o_ego; }
    else if (q == 3) { v = clipf((speed_kmh(ag.v) + 1.0f) / (max_speed + 1.0f), 0.0f, 1.0f); col = o_ego + 1; }
    else if (q == 4) { v = clipf((ag.steer / 60.0f + 1.0f) * 0.5f, 0.0f, 1.0f); col = o_ego + 2; }
    else if (q == 5) { v = clipf((ag.a0s + 1.0f) * 0.5f, 0.0f, 1.0f); col = o_ego + 3; }
    else if (q == 6) { v = clipf((ag.a0t + 1.0f) * 0.5f, 0.0f, 1.0f); col = o_ego + 4; }
    else if (q == 7) {
      // acos(clip(cos_beta, 0, 1)) (state_obs.py:87-92) evaluated as atan2(|cross|, dot): identical for unit vectors,
      // but well-conditioned in fp32 near beta = 0 where 1 - cos(beta) underflows the mantissa
      // ... and only beta <= 0.1 rad survives the clip below: there atan(z) = z - z^3 / 3 + z^5 / 5 to 1.4e-8 (z <= 0.1004), six
      // instructions instead of the library atan2f's 55; anything larger is 1 after the clip whatever its exact value
      float dot = hx * ag.lhx + hy * ag.lhy, cross = hx * ag.lhy - hy * ag.lhx;
      const float z = dot > 0.0f ? fabsf(cross) / dot : 1.0f, q = z * z;
      float beta = z > 0.11f ? 1.0f : z * (1.0f + q * (-1.0f / 3.0f + q * 0.2f));
      v = clipf(beta / 0.1f, 0.0f, 1.0f);
      col = o_ego + 5;
    } else {  // lanes 8..12 -> checkpoint 1, 13..17 -> checkpoint 2
      int which = (q - 8) / 5, comp = (q - 8) - which * 5;
      float out[5];
      navi_info_for(nv, mv.lane_width(), ag.cur_n, px, py, hx, hy, out);
      v = comp == 0 ? out[0] : comp == 1 ? out[1] : comp == 2 ? out[2] : comp == 3 ? out[3] : out[4];
      col = toll ? -1 : o_navi + (q - 8);
    }
    if (col >= 0) OBS_ST(row + col, v);
  }
  // SideDetector / LaneLineDetector fans (distance_detector.py:137-152): beam i at theta + i*2pi/n + 90 deg, cast through
  // the map grid against the line boxes of the wanted kinds
  for (int q = tid; q < KS + KM; q += nt) {
    const bool side = q < KS;
    const int i = side ? q : q - KS, n = side ? KS : KM;
    const float dist = side ? d.cfg.side_dist : d.cfg.lane_line_dist;
    const unsigned kinds = side ? ((1u << PGD_BOX_WHITE) | (1u << PGD_BOX_YELLOW))
                                : ((1u << PGD_BOX_WHITE) | (1u << PGD_BOX_YELLOW) | (1u << PGD_BOX_BROKEN));
    float sn, cs;
    sincosf((float)i * (2.0f * PGD_PI / (float)n) + 0.5f * PGD_PI + ag.th, &sn, &cs);
    OBS_ST(row + (side ? i : o_ego + 6 + i), ray_grid(mv, px, py, dist * cs, dist * sn, kinds));
  }
}

// state_block for ONE thread and the plain row layout -- [2 lateral distances][6 ego floats][10 navigation floats], no detector fans,
// no random-agent-model floats, no toll floats: the three table reads at once, the eighteen floats straight down (state_block's loop
// would run its branch ladder once per float), nine 8-byte stores (a row starts at a multiple of D floats: D even -> 8-byte aligned,
// else single stores).  Same expressions as state_block, float by float.
template <class MV>
DEV void state_block_one(const PgdDev& d, const MV& mv, const pgd_spawn& sp, const AgentView& ag, float* __restrict__ row) {
  const float px = ag.x, py = ag.y, hx = ag.hx, hy = ag.hy;
  const pgd_lane ml = mv.lanes[state_lane_of(ag, 2)];
  const LaneNav nv0 = mv.lnav()[state_lane_of(ag, 8)], nv1 = mv.lnav()[state_lane_of(ag, 13)];
  float o[18];
  o[0] = clipf(ag.dl / 18.0f, 0.0f, 1.0f);
  o[1] = clipf(ag.dr / 18.0f, 0.0f, 1.0f);
  o[2] = heading_diff(ml, px, py, hx, hy);
  o[3] = clipf((speed_kmh(ag.v) + 1.0f) / (sp.max_speed + 1.0f), 0.0f, 1.0f);
  o[4] = clipf((ag.steer / 60.0f + 1.0f) * 0.5f, 0.0f, 1.0f);
  o[5] = clipf((ag.a0s + 1.0f) * 0.5f, 0.0f, 1.0f);
  o[6] = clipf((ag.a0t + 1.0f) * 0.5f, 0.0f, 1.0f);
  {
    const float dot = hx * ag.lhx + hy * ag.lhy, cross = hx * ag.lhy - hy * ag.lhx;
    const float z = dot > 0.0f ? fabsf(cross) / dot : 1.0f, q = z * z;
    const float beta = z > 0.11f ? 1.0f : z * (1.0f + q * (-1.0f / 3.0f + q * 0.2f));
    o[7] = clipf(beta / 0.1f, 0.0f, 1.0f);
  }
  navi_info_for(nv0, mv.lane_width(), ag.cur_n, px, py, hx, hy, o + 8);
  navi_info_for(nv1, mv.lane_width(), ag.cur_n, px, py, hx, hy, o + 13);
  // 8-byte stores only where every row starts on an 8-byte boundary: even row width AND even row stride (pgd_step_packed takes any
  // stride >= A * (D + 2)) AND an aligned buffer -- the same guard as the lidar fan's pairs_ok
  if (((d.D | d.ostride) & 1) == 0 && (reinterpret_cast<uintptr_t>(row) & 7) == 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) reinterpret_cast<float2*>(row)[k] = make_float2(o[2 * k], o[2 * k + 1]);
  } else {
#pragma unroll
    for (int k = 0; k < 18; ++k) row[k] = o[k];
  }
}

// the view of slot `o` of the env as an observed vehicle (the state vector a neighbour contributes to
// LidarStateObservationMARound); its speed comes from the observer's snapshot
DEV AgentView view_of_slot(const PgdDev& d, const MapView& mv, const RecPiece* recs, const pgd_spawn* spb, int o, float spd_kmh,
                           int env, uint32_t tick) {
  Veh rc;
  load_rec(recs, d.V, o, rc);
  AgentView ag;
  ag.x = rc.x; ag.y = rc.y; ag.th = rc.th;
  ag.hx = rc.hx; ag.hy = rc.hy;
  ag.dl = rc.dl; ag.dr = rc.dr;
  ag.v = spd_kmh == 0.0f ? 0.0f : rc.v;  // the snapshot says 0: an agent that finished in an EARLIER step (static body)
  ag.steer = rc.steer; ag.a0s = rc.a0s; ag.a0t = rc.a0t; ag.lhx = rc.lasthx; ag.lhy = rc.lasthy;
  ag.cur_first = rc.cur_first; ag.cur_n = rc.cur_n; ag.next_first = rc.next_first;
  ag.blk = rc.blk; ag.toll_time = rc.php;
  ag.env = env; ag.slot = o; ag.tick = tick;
  return ag;
}

// writes the D floats of one agent's row with `nt` cooperating threads (tid in [0, nt))
// STD: the reference's default row layout (no detector fans, no random_agent_model, no toll floats, no lidar noise) as a
// compile-time fact: every column offset is a constant and the optional blocks vanish from the benchmark kernel.
// OTH: PGD_MA_OTHERS_STATE (stand-alone k_observe only: `recs` / `spb` = the env's records and spawn table)
// gaussian noise / dropout of one lidar value (state_obs.py:172-182), from the counter RNG keyed by (env, agent slot, beam, step)
DEV float lidar_noise(const PgdDev& d, int env, int slot, uint32_t tick, int i, float best) {
  if (!(d.cfg.lidar_gaussian_noise > 0.0f || d.cfg.lidar_dropout_prob > 0.0f)) return best;
  const uint32_t key = 0x51d0a000u + (uint32_t)slot * 1024u + (uint32_t)i;
  if (d.cfg.lidar_gaussian_noise > 0.0f) {
    const float u1 = ((float)(pgd_rng(d.cfg.seed, (uint32_t)(d.cfg.env_base + env), key, tick) >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(pgd_rng(d.cfg.seed ^ 0x9e3779b9u, (uint32_t)(d.cfg.env_base + env), key, tick) >> 8) + 0.5f) * (1.0f / 16777216.0f);
    best = clipf(best + d.cfg.lidar_gaussian_noise * sqrtf(-2.0f * logf(u1)) * cosf(2.0f * PGD_PI * u2), 0.0f, 1.0f);
  }
  if (d.cfg.lidar_dropout_prob > 0.0f) {
    const float u3 = ((float)(pgd_rng(d.cfg.seed ^ 0x7f4a7c15u, (uint32_t)(d.cfg.env_base + env), key, tick) >> 8) + 0.5f) * (1.0f / 16777216.0f);
    if (u3 < d.cfg.lidar_dropout_prob) best = 0.0f;
  }
  return best;
}

// the threads that produce one row meet: a block-wide barrier, or -- when the row belongs to ONE wave of a block that holds
// several rows -- nothing but the ordering of that wave's own LDS traffic (the waves of the block stay independent)
template <bool WAVE_ROW>
DEV void row_sync() {
  if (WAVE_ROW) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } else __syncthreads();
}

// STATE = false: the caller has written the state block (and the toll floats' inputs are unchanged) already
// WAVE_ROW: see row_sync
template <bool OBJ, bool STD = false, bool OTH = false, bool STATE = true, bool WAVE_ROW = false, class MV>
DEV void observe_agent(const PgdDev& d, const MV& mv, const pgd_spawn& sp, const AgentView& ag, ObsLds& L,
                       float* __restrict__ row, int tid, int nt, const RecPiece* recs = nullptr, const pgd_spawn* spb = nullptr,
                       const ObsPre* pre = nullptr, unsigned* minb = nullptr) {
  const float px = ag.x, py = ag.y, hx = ag.hx, hy = ag.hy;
  const float R = d.cfg.lidar_dist;
  const int NL = d.cfg.num_lasers;
  const int KS = STD ? 0 : d.cfg.side_lasers, KM = STD ? 0 : d.cfg.lane_line_lasers;
  const bool toll = !STD && (d.cfg.marl_flags & PGD_MA_TOLLGATE) != 0;
  const int RAM = (!STD && d.cfg.random_agent_model) ? 2 : 0;
  const int o_oth = (KS > 0 ? KS : 2) + 6 + KM + RAM + (toll ? 0 : 10);  // = length of the state block
  const int NO = d.cfg.num_others;
  const int per_other = OTH ? o_oth : 4;
  // the beam directions of this thread's first four rounds (the whole fan when NL <= 4 nt: 240 beams, one wave): read before the
  // state block (or earlier still, by the caller: obs_preload); read inside the round loop each one is waited for on the spot,
  // behind the previous round's row store
  float2 bd0 = make_float2(0.0f, 0.0f), bd1 = bd0, bd2 = bd0, bd3 = bd0;
  if (pre) { bd0 = pre->bd0; bd1 = pre->bd1; bd2 = pre->bd2; bd3 = pre->bd3; }
  else if (NL > 0) {
    bd0 = d.beam[min(tid, NL - 1)]; bd1 = d.beam[min(nt + tid, NL - 1)];
    bd2 = d.beam[min(2 * nt + tid, NL - 1)]; bd3 = d.beam[min(3 * nt + tid, NL - 1)];
  }
  if (STATE) state_block<STD>(d, mv, sp, ag, row, tid, nt, pre);
  if (toll && tid == 0) {  // TollGateObservation.observe (marl_tollgate.py:84-96)
    const bool in_toll = ag.blk == '$';
    float* t2 = row + o_oth + per_other * NO + NL;
    t2[0] = in_toll ? 1.0f : 0.0f;
    t2[1] = (in_toll && ag.toll_time > (float)d.cfg.min_pass_steps) ? 1.0f : 0.0f;
  }
  PHASE_MARK(22);  // obs: state + navi block
  XMARK(22);
  if (NL <= 0) return;
  // get_surrounding_vehicles_info (lidar.py:55-77): rank by centre distance (stable), 4 floats per neighbour; the last
  // threads take this part so that it overlaps the state block of the first ones
  const int n = L.n, nveh = OBJ ? L.nveh : n;
  if (OTH) {
    // LidarStateObservationMARound.observe (marl_inout_roundabout.py:82-105): the num_others nearest vehicles (stable rank
    // by centre distance) contribute their own state vectors; absent ranks are zeros.  Ranks -> slots through LDS, then
    // the block evaluates one neighbour after the other with the same state_block code.
    if (tid < 16) L.rank_slot[tid] = -1;
    row_sync<WAVE_ROW>();
    for (int k = tid; k < n; k += nt) {
      int rank = 0;
      const float dk = L.bdist[k];
      for (int j = 0; j < n; ++j) rank += (L.bdist[j] < dk || (L.bdist[j] == dk && j < k)) ? 1 : 0;
      if (rank < NO && dk < __builtin_inff()) { L.rank_slot[rank] = L.bslot[k]; L.rank_spd[rank] = L.bspd[k]; }
    }
    row_sync<WAVE_ROW>();
    for (int r = 0; r < NO; ++r) {
      const int o = L.rank_slot[r];
      float* dst = row + o_oth + r * o_oth;
      if (o < 0) {
        for (int q = tid; q < o_oth; q += nt) dst[q] = 0.0f;
      } else {
        const AgentView oa = view_of_slot(d, mv, recs, spb, o, L.rank_spd[r], ag.env, ag.tick);
        state_block<STD>(d, mv, spb[rec_spawn(recs, d.V, o)], oa, dst, tid, nt);
      }
    }
  } else
  // with objects: indices [0, n) are the compacted bodies, [n, n + NO) the rank rows to zero-fill; without: [0, max(n, NO))
  for (int k = nt - 1 - tid; k < (OBJ ? n + NO : (n > NO ? n : NO)); k += nt) {
    if (k < n) {
      int rank = 0;
      float dk = L.bdist[k];
      for (int j = 0; j < n; ++j) rank += (L.bdist[j] < dk || (L.bdist[j] == dk && j < k)) ? 1 : 0;
      if (rank < NO && dk < __builtin_inff()) {
        float ph, ps;
        float ms = sp.max_speed;
        float sp_me = speed_kmh(ag.v);
        projection(hx, hy, L.bx[k] - px, L.by[k] - py, ph, ps);
        float* o = row + o_oth + rank * 4;
        o[0] = clipf((ph / R + 1.0f) * 0.5f, 0.0f, 1.0f);
        o[1] = clipf((ps / R + 1.0f) * 0.5f, 0.0f, 1.0f);
        projection(hx, hy, L.bspd[k] * L.bux[k] - sp_me * hx, L.bspd[k] * L.buy[k] - sp_me * hy, ph, ps);
        o[2] = clipf((ph / ms + 1.0f) * 0.5f, 0.0f, 1.0f);
        o[3] = clipf((ps / ms + 1.0f) * 0.5f, 0.0f, 1.0f);
      }
    } else if (!OBJ || k - n >= nveh) {  // ranks [nveh, NO): absent neighbour -> zeros
      float* o = row + o_oth + (OBJ ? k - n : k) * 4;
      o[0] = o[1] = o[2] = o[3] = 0.0f;
    }
  }
  PHASE_MARK(23);  // obs: neighbours
  XMARK(23);
  // lidar (distance_detector.py:65-94, cutils.pyx:60-142): beam i at theta + i*2pi/N, nearest hit fraction
  // every lane of the row takes part in every round (the ballot below needs the body lanes), beams past the fan are not stored
  const int lane64 = tid & (WAVE - 1);
  const unsigned my_sec = L.bsec[lane64 < n ? lane64 : 0];  // sectors reached by the body this lane stands for (same in every round)
  auto cast_round = [&](const int i0, const float2 bd) {  // bd = (cos, sin)(i * 2 pi / NL); rotated by the heading
    const int i = i0 + tid;
    const bool on = i < NL;
    const float dx = R * (bd.x * hx - bd.y * hy), dy = R * (bd.y * hx + bd.x * hy);
    float best = 1.0f;
    // the beams a wave casts in one round lie in one sector of 64 (nt is a multiple of 64): bodies whose window misses the sector
    // are skipped for the whole wave -- a scalar walk over the set bits of a ballot -- the others take the per-beam window test
    const int sec = (i0 + (tid & ~(WAVE - 1))) >> 6;
    unsigned long long todo = __ballot(lane64 < n && ((my_sec >> (sec & 31)) & 1u) != 0u);
    for (; todo != 0ull; todo &= todo - 1ull) {
      const int k = __builtin_ctzll(todo);
      // window and box of body k in one round trip (the box is needed by some beam of the round: that is what the sector bit says)
      const int bi0 = L.bi0[k], bcnt = L.bcnt[k];
      const Obb box{L.bx[k], L.by[k], L.bux[k], L.buy[k], L.bhl[k], L.bhw[k]};
      int off = i - bi0;
      off += off < 0 ? NL : 0;
      if (on && off < bcnt) best = fminf(best, shape_ray<OBJ>(box, px, py, dx, dy));
    }
    if (!STD) best = lidar_noise(d, ag.env, ag.slot, ag.tick, i, best);
    if (on) OBS_ST(row + o_oth + per_other * NO + i, best);
  };
  // With several bodies in range a round of 64 beams runs the slab test of every body whose window meets its 96-degree sector for
  // ALL 64 beams, although a window is 10 - 20 beams wide: three quarters of those tests are masked off.  A row that one wave
  // produces and whose fan fits the four direction registers (240 beams: the default) instead flattens the (body, beam-inside-its-
  // window) INCIDENCES by a prefix sum over the window sizes and deals them out to the lanes 64 at a time: a body is tested against
  // the beams that can reach it and nothing else; the nearest hit per beam is an unsigned min in LDS (fractions are >= 0, so
  // their bit patterns order like the floats).  Same pairs tested as by the rounds below, same minimum: the same cloud.
  // `minb` = NL words of LDS nobody else uses during the row.  Wave-uniform switch: a single body is cheaper in the round loop.
  if (WAVE_ROW && minb != nullptr && nt == WAVE && NL <= 4 * WAVE) {
    const int n_u = __builtin_amdgcn_readfirstlane(n);
    if (n_u >= PGD_LIDAR_INC_MIN) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q * WAVE + lane64 < NL) minb[q * WAVE + lane64] = 0x3f800000u;  // 1.0f = no hit
      // exclusive prefix of the window sizes: lane k < n stands for body k
      const int cnt = lane64 < n_u ? L.bcnt[lane64] : 0;
      int incl = cnt;
      for (int off = 1; off < n_u; off <<= 1) {
        const int up = __shfl_up(incl, off);
        incl += lane64 >= off ? up : 0;
      }
      const int pref = incl - cnt;
      const int total = __builtin_amdgcn_readlane(incl, n_u - 1);
      const int rounds = (total + WAVE - 1) / WAVE;
      // per round of 64 incidences a 64-bit mask of the positions at which a body's window starts (behind the minima, 8-byte
      // aligned): the owner of an incidence is then (windows started in earlier rounds) + (starts at or below its position) - 1
      unsigned long long* starts = reinterpret_cast<unsigned long long*>(minb + ((NL + 1) & ~1));
      const int max_rounds = (PGD_LIDAR_MINB_WORDS - ((NL + 1) & ~1)) / 2;
      if (rounds <= max_rounds) {
      if (lane64 < rounds) starts[lane64] = 0ull;
      if (lane64 < n_u) L.bpref[lane64] = pref;
      // this lane's own four beams, rotated by the heading: an incidence fetches the direction of its beam from the lane that holds it
      const float dx0 = R * (bd0.x * hx - bd0.y * hy), dy0 = R * (bd0.y * hx + bd0.x * hy);
      const float dx1 = R * (bd1.x * hx - bd1.y * hy), dy1 = R * (bd1.y * hx + bd1.x * hy);
      const float dx2 = R * (bd2.x * hx - bd2.y * hy), dy2 = R * (bd2.y * hx + bd2.x * hy);
      const float dx3 = R * (bd3.x * hx - bd3.y * hy), dy3 = R * (bd3.y * hx + bd3.x * hy);
      row_sync<true>();
      if (lane64 < n_u) atomicOr(&starts[pref >> 6], 1ull << (pref & 63));
      row_sync<true>();
      const int my_round = lane64 < n_u ? (pref >> 6) : 0x7fffffff;
      for (int rd = 0; rd < rounds; ++rd) {
        const int j = rd * WAVE + lane64;
        const unsigned long long sm = starts[rd];
        const int before = __popcll(__ballot(my_round < rd));
        const int owner = before + __popcll(sm & ((2ull << lane64) - 1ull)) - 1;
        const bool on = j < total;
        const int k = on ? owner : 0;
        int i = L.bi0[k] + (j - L.bpref[k]);
        i -= i >= NL ? NL : 0;
        i = on ? i : 0;
        const Obb box{L.bx[k], L.by[k], L.bux[k], L.buy[k], L.bhl[k], L.bhw[k]};
        const int src = i & (WAVE - 1), q = i >> 6;
        float dx = __shfl(dx0, src), dy = __shfl(dy0, src);
        if (NL > WAVE) { const float a = __shfl(dx1, src), b = __shfl(dy1, src); dx = q == 1 ? a : dx; dy = q == 1 ? b : dy; }
        if (NL > 2 * WAVE) { const float a = __shfl(dx2, src), b = __shfl(dy2, src); dx = q == 2 ? a : dx; dy = q == 2 ? b : dy; }
        if (NL > 3 * WAVE) { const float a = __shfl(dx3, src), b = __shfl(dy3, src); dx = q == 3 ? a : dx; dy = q == 3 ? b : dy; }
        const float t = shape_ray<OBJ>(box, px, py, dx, dy);
        if (on && t < 1.0f) atomicMin(&minb[i], __float_as_uint(t));
      }
      row_sync<true>();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = q * WAVE + lane64;
        if (i < NL) {
          float best = __uint_as_float(minb[i]);
          if (!STD) best = lidar_noise(d, ag.env, ag.slot, ag.tick, i, best);
          OBS_ST(row + o_oth + per_other * NO + i, best);
        }
      }
      PHASE_MARK(24);  // obs: lidar
      return;
      }
    }
  }
  if (0 < NL) cast_round(0, bd0);
  if (nt < NL) cast_round(nt, bd1);
  if (2 * nt < NL) cast_round(2 * nt, bd2);
  if (3 * nt < NL) cast_round(3 * nt, bd3);
  for (int i0 = 4 * nt; i0 < NL; i0 += nt) cast_round(i0, d.beam[min(i0 + tid, NL - 1)]);
  PHASE_MARK(24);  // obs: lidar
}

// ---------------------------------------------------------------------------------------------------------------------
// observe_env_body (kernel k_observe_env, and the tail of the multi-agent k_step): the rows of ALL agents of an env by one wave (multi-agent engines; same results as k_observe, row by row).
// A block per row spends its life waiting for a handful of loads, 8 x A of them per env.  Here the env's records are read
// once, and the work is laid out by what there is to do instead of by row:
//   state blocks   WAVE / A lanes per agent, every agent at once (state_block with few threads);
//   pairs          lane = (observer, body): broad phase, beam window, neighbour rank -- WAVE / V observers per pass;
//   lidar          the (observer, body, beam-inside-the-window) incidences of the pass, flattened by a prefix sum over the
//                  pairs' window sizes and dealt out to the lanes 64 at a time: a body is tested against the few beams that
//                  can reach it and nothing else; the nearest hit per beam is an unsigned min in LDS (fractions are >= 0).
// ---------------------------------------------------------------------------------------------------------------------
// NW waves per env: the state blocks get NW * WAVE / A lanes per agent and the passes of the pair phase are dealt out to the waves
// (wave w takes passes w, w + NW, ...; each wave has its own scratch and synchronises with itself only).  NW = 4 when there
// are at least four passes (A >= 4 * (WAVE / V)), else 1.
// NW waves per env: the state blocks get NW * WAVE / A lanes per agent and the passes of the pair phase are dealt out to the waves
// (wave w takes passes w, w + NW, ...; each wave has its own scratch and synchronises with itself only).
// LDS words one wave needs for rounds of g observers
// per observer of a round: the per-beam minima, the candidate pairs (two to a word), and -- engines that observe neighbour rows --
// centre distance and speed of every pair
DEV_HOST int observe_env_words(int g, int num_lasers, int V, int num_others) {
  return g * ((num_lasers > 0 ? num_lasers : 0) + (V + 1) / 2 + (num_others > 0 ? 2 * V : 0));
}
// ... and behind the waves' areas, with PGD_MA_OTHERS_STATE: slot and speed of every observer's ranked neighbours
DEV_HOST int observe_env_oth_words(int A, int num_others, bool oth) { return oth ? 2 * A * num_others : 0; }

// CAP: slots the body / observer tables hold (WAVE, or the seat count rounded up to 16 in the instantiations that fold it: with 48
// instead of 64 and the beam start packed into the pair word a block of the 40-seat kernel takes 20.2 instead of 22.0 KB, and the
// eighth block fits the CU's 160 KB)
template <int NW, int CAP = WAVE>
struct ObsEnvLds {
  float bX[CAP], bY[CAP], bUX[CAP], bUY[CAP], bHL[CAP], bHW[CAP], bV[CAP], bAID[CAP];
  int bST[CAP];       // status | kind << 8
  uint32_t bFL[CAP];  // step flags of agent slot o (0 beyond A or without flags)
  float aMS[CAP];     // observer: max_speed of its vehicle
  int aWant[CAP];
  unsigned char wList[CAP], bList[CAP];  // the observers that get a row / the bodies that can be seen by one, ascending
  float pDist[NW][WAVE];  // (as ints: first beam of the window << 16 | observer of the round << 8 | body, per lane of the pass)
  int pPref[NW][WAVE + 1];
  unsigned long long pMask[NW][32];  // per round of 64 incidences: the positions at which a pair's window starts
};
// `G` observers per round of a wave (as many as the LDS holds, see observe_env_words);
// `s_minb_all`: per wave G * (num_lasers + 2 V) words of LDS: nearest hit fraction per (observer of the round, beam) as float
// bits, then centre distance and speed of every pair of the round
// ALLOW_OTH = false compiles the neighbour-state-vector phase (PGD_MA_OTHERS_STATE) out: the copy appended to k_step stays as small
// as it was (every addition to that kernel costs SGPR spills across the whole step)
// What the wave that has just stepped the env still holds of it (k_step's multi-agent tail, engines without traffic slots: the
// lanes of agent `slot` are exactly the lanes observe_env_body gives its state block): with it the routine starts from registers
// instead of reading back from memory the records, flags, env words, map header and spawn records the wave has just written or used.
struct EnvInWave {
  const Veh* me;          // this lane's vehicle
  const pgd_spawn* sp;    // its spawn record (scalar head)
  const MapView* mv;
  uint32_t fl;            // its step flags
  int scen;
  uint32_t tick;
  int slot, sub;
};
// x / n for 0 <= x < 4096, n <= 128 without the integer division (~25 instructions): (x + 0.5) / n is at least 0.5 / n away from an
// integer and the fp32 product is off by < 1e-3 of one
DEV int div_small(const int x, const float inv_n) { return (int)(((float)x + 0.5f) * inv_n); }
// identity of an observation buffer for the zero-row marks (PgdDev::rowz): its address mixed with the row stride
DEV unsigned long long rowz_tag(const float* obs, int ostride) {
  return (unsigned long long)(uintptr_t)obs ^ ((unsigned long long)(unsigned)ostride * 0x9E3779B97F4A7C15ull);
}
// OBJ = false: engines whose scenarios hold no traffic objects (no circles among the bodies)
// STATE = false: the state blocks of the rows that are due have been written by k_step (PgdDev::state_rows): no record of an
// observer, no spawn record, no lane table is read here -- the bodies' poses and the step flags are all the routine needs
template <int NW, bool ALLOW_OTH = true, bool FUSED = false, bool OBJ = true, bool STATE = true, int CAP = WAVE>
DEV void observe_env_body(const PgdDev& d, int e, float* __restrict__ obs, const uint32_t* flags, ObsEnvLds<NW, CAP>& M,
                          unsigned* s_minb_all, const int G, const EnvInWave* in_wave = nullptr) {
  float (&bX)[CAP] = M.bX; float (&bY)[CAP] = M.bY; float (&bUX)[CAP] = M.bUX; float (&bUY)[CAP] = M.bUY;
  float (&bHL)[CAP] = M.bHL; float (&bHW)[CAP] = M.bHW; float (&bV)[CAP] = M.bV; float (&bAID)[CAP] = M.bAID;
  int (&bST)[CAP] = M.bST; uint32_t (&bFL)[CAP] = M.bFL; float (&aMS)[CAP] = M.aMS; int (&aWant)[CAP] = M.aWant;
  float (&pDist_all)[NW][WAVE] = M.pDist; int (&pPref_all)[NW][WAVE + 1] = M.pPref;
  const int V = d.V, A = d.A, D = d.D, NL = d.cfg.num_lasers, NO = d.cfg.num_others;
  const int tid = threadIdx.x, wv = tid / WAVE, lane = tid % WAVE;
  const RecPiece* recs = rec_block(d.rec, (size_t)e, V);
  // ---- loads whose addresses follow from the block index: body `lane` (first half of its record), the agent of this lane's
  // state group (whole record), step flags, scenario, step count, the env's map header
  const int LPA = WAVE * NW / A;  // lanes per agent in the state phase (A <= WAVE)
  const int sa = FUSED ? in_wave->slot : tid / LPA, st = FUSED ? in_wave->sub : tid - (tid / LPA) * LPA;
  const bool s_on = sa < A;
  // (read by every wave before the first barrier, written behind it)
  const unsigned long long rz_tag = rowz_tag(obs, d.ostride);
  const ulonglong2 rz = d.rowz ? d.rowz[e] : make_ulonglong2(0ull, 0ull);
  const unsigned long long known_zero = rz.y == rz_tag ? rz.x : 0ull;  // marks of another buffer say nothing about this one
  Veh me;
  Veh body;  // only the first 64 bytes are filled
  uint32_t f_me, f_body = 0u;
  int scen;
  uint32_t tick;
  MapView mv;
  if (FUSED) {
    me = *in_wave->me;
    f_me = s_on ? in_wave->fl : 0u;
    scen = in_wave->scen; tick = in_wave->tick; mv = *in_wave->mv;
  } else {
    if (STATE) load_rec(recs, V, s_on ? sa : 0, me);
    load_rec_head(recs, V, tid < V ? tid : 0, body);
    f_me = (STATE && flags && s_on) ? flags[(size_t)e * A + sa] : 0u;
    f_body = (flags && tid < A) ? flags[(size_t)e * A + tid] : 0u;
    scen = d.ei[(size_t)(e) * PGD_NEI + EI_SCEN];
    tick = (uint32_t)d.ei[(size_t)(e) * PGD_NEI + EI_STEPS_TOTAL];
    if (STATE) mv = map_view_of(d, d.env_map + e);
  }
  const pgd_spawn* spb = d.spawns + (size_t)scen * d.sstride;
  const pgd_spawn& msp = FUSED ? *in_wave->sp : spb[STATE ? me.spawn : 0];
  // which slots get a row: after a multi-agent step the ones that reported or were (re)spawned, else the active ones
  bool want = false;  // (STATE = false: the rows' state blocks are there already; `aWant` comes from the bodies' own lanes below)
  if (STATE) {
    want = me.status == ST_ACTIVE;
    if (flags) want = (f_me & PGD_F_RESET) ? want : (f_me & (PGD_F_REPORT | PGD_F_NEW)) != 0;
    want = want && s_on;
  }
  // ---- publish the bodies and the observers
  if (FUSED) {  // V == A: the first lane of every agent publishes its own vehicle
    if (s_on && st == 0) {
      const int so_kind = msp.kind;
      bX[sa] = me.x; bY[sa] = me.y; bUX[sa] = me.hx; bUY[sa] = me.hy;
      bHL[sa] = 0.5f * msp.length; bHW[sa] = so_kind == PGD_OBJ_CYLINDER ? -1.0f : 0.5f * msp.width;
      bV[sa] = me.v; bAID[sa] = me.agent_id;
      bST[sa] = (int)me.status | (so_kind << 8);
      bFL[sa] = f_me;
    }
  } else if (tid < V) {
    // (one body size for the whole engine -- PgdDev::uni_len, a kernel argument: no spawn record is read, the publish follows the
    // records' arrival directly; a wave-uniform branch)
    float so_len, so_wid;
    int so_kind;
    if (!OBJ && d.uni_len > 0.0f) { so_len = d.uni_len; so_wid = d.uni_wid; so_kind = PGD_OBJ_VEHICLE; }
    else {
      const pgd_spawn& so = spb[body.spawn];
      so_len = so.length; so_wid = so.width; so_kind = so.kind;
    }
    bX[tid] = body.x; bY[tid] = body.y; bUX[tid] = body.hx; bUY[tid] = body.hy;
    bHL[tid] = 0.5f * so_len; bHW[tid] = so_kind == PGD_OBJ_CYLINDER ? -1.0f : 0.5f * so_wid;
    bV[tid] = body.v; bAID[tid] = body.agent_id;
    bST[tid] = (int)body.status | (so_kind << 8);
    bFL[tid] = f_body;
  }
  if (STATE) {
    if (s_on && st == 0) { aMS[sa] = msp.max_speed; aWant[sa] = want ? 1 : 0; }
  } else if (tid < A) {  // the same rule from the body's own status and flags (its lane holds both)
    bool w = body.status == ST_ACTIVE;
    if (flags) w = (f_body & PGD_F_RESET) ? w : (f_body & (PGD_F_REPORT | PGD_F_NEW)) != 0;
    aWant[tid] = w ? 1 : 0;
    aMS[tid] = 0.0f;  // (only read for neighbour rows, which the STATE = false kernels do not write)
  }
  // ---- state blocks: every agent at once, LPA lanes each
  float* row = obs + (size_t)e * d.ostride + (size_t)(s_on ? sa : 0) * D;
  if (want) {
    AgentView ag;
    ag.x = me.x; ag.y = me.y; ag.th = me.th;
    ag.hx = me.hx; ag.hy = me.hy;
    ag.dl = me.dl; ag.dr = me.dr; ag.v = me.v; ag.steer = me.steer;
    ag.a0s = me.a0s; ag.a0t = me.a0t; ag.lhx = me.lasthx; ag.lhy = me.lasthy;
    ag.cur_first = me.cur_first; ag.cur_n = me.cur_n; ag.next_first = me.next_first;
    ag.blk = me.blk; ag.toll_time = me.php;
    ag.env = e; ag.slot = sa; ag.tick = tick;
    state_block<false>(d, mv, msp, ag, row, st, LPA);
    if ((d.cfg.marl_flags & PGD_MA_TOLLGATE) && st == 0) {  // TollGateObservation.observe (marl_tollgate.py:84-96)
      const int KS = d.cfg.side_lasers, KM = d.cfg.lane_line_lasers, RAM = d.cfg.random_agent_model ? 2 : 0;
      const bool in_toll = ag.blk == '$';
      const int sl = (KS > 0 ? KS : 2) + 6 + KM + RAM;  // state block of the tollgate row (no navigation floats)
      const bool oth_s = ALLOW_OTH && (d.cfg.marl_flags & PGD_MA_OTHERS_STATE) != 0 && NO > 0;
      float* t2 = row + sl + (oth_s ? sl : 4) * NO + NL;
      t2[0] = in_toll ? 1.0f : 0.0f;
      t2[1] = (in_toll && ag.toll_time > (float)d.cfg.min_pass_steps) ? 1.0f : 0.0f;
    }
  }
  __syncthreads();
  // ---- who observes, what can be seen.  A multi-agent env has as many slots as agents may ever be alive at once (40 on the
  // reference's roundabout) and, under most policies, far fewer agents in the world: the rows that are due and the bodies that can
  // show up in one (in the world now, or -- a row of the step shows the world before the step's finishes -- reported this step)
  // are compacted into two ascending lists, and everything below works on observers x bodies of the LISTS, 64 pairs at a time.
  // (Every wave builds the same lists: identical bytes to identical places.)
  unsigned char* wList = M.wList; unsigned char* bList = M.bList;
  const int tl = (CAP < WAVE && lane >= CAP) ? 0 : lane;  // (lanes beyond the tables: V, A <= CAP, their verdicts are false anyway)
  const bool w_l = lane < A && aWant[tl] != 0;
  const int st_l = bST[tl] & 0xff;
  const bool b_l = lane < V && (st_l == ST_PENDING || st_l == ST_ACTIVE || st_l == ST_DYING || (lane < A && (bFL[tl] & PGD_F_REPORT) != 0u));
  const unsigned long long wm = __ballot(w_l), bm = __ballot(b_l);
  const int nW = __popcll(wm), nB = __popcll(bm);
  if (w_l) wList[__popcll(wm & ((1ull << lane) - 1ull))] = (unsigned char)lane;
  if (b_l) bList[__popcll(bm & ((1ull << lane) - 1ull))] = (unsigned char)lane;
  // rows that are not due read zero: written by the whole block, one row after the other (coalesced), not by the slot's own lanes
  // -- and only the rows that are not known to be zero already (PgdDev::rowz)
  const unsigned long long not_due = (A >= 64 ? ~0ull : ((1ull << A) - 1ull)) & ~wm;
  for (unsigned long long zm = not_due & ~known_zero; zm != 0ull; zm &= zm - 1ull) {
    float* zr = obs + (size_t)e * d.ostride + (size_t)__builtin_ctzll(zm) * D;
    for (int k = tid; k < D; k += WAVE * NW) zr[k] = 0.0f;
  }
  if (d.rowz && tid == 0 && (not_due != known_zero || rz.y != rz_tag)) d.rowz[e] = make_ulonglong2(not_due, rz_tag);
  row_sync<true>();
  PHASE_MARK(20);  // env obs: loads, publish, state blocks, lists, zero rows
  if (NL <= 0) return;
  // PGD_MA_OTHERS_STATE (LidarStateObservationMARound): a neighbour row is the neighbour's own state vector; the ranks found by
  // the pair phase are parked in LDS (slot, speed as the observer sees it) and the vectors are written by a last phase below
  const bool oth = ALLOW_OTH && (d.cfg.marl_flags & PGD_MA_OTHERS_STATE) != 0 && NO > 0;
  const size_t wave_words = (size_t)observe_env_words(G, NL, V, NO);
  int* nbSlot = reinterpret_cast<int*>(s_minb_all + (size_t)NW * wave_words);
  float* nbSpd = reinterpret_cast<float*>(nbSlot + A * NO);
  if (oth) {
    for (int k = tid; k < A * NO; k += WAVE * NW) nbSlot[k] = -1;
    __syncthreads();
  }
  // ---- pairs.  Every wave owns a contiguous range of observers and works through it in rounds of at most `G` observers (what
  // the LDS for the per-beam minima holds); the (observer, body) pairs of a round are packed into the lanes 64 at a time,
  // whatever V is (V = 40: 25 passes for 40 observers instead of 40 passes with 40 busy lanes each).
  const bool toll = (d.cfg.marl_flags & PGD_MA_TOLLGATE) != 0;
  const int o_oth = (d.cfg.side_lasers > 0 ? d.cfg.side_lasers : 2) + 6 + d.cfg.lane_line_lasers + (d.cfg.random_agent_model ? 2 : 0) + (toll ? 0 : 10);
  const float R = d.cfg.lidar_dist, R_lidar = R;
  const int per_wave = (nW + NW - 1) / NW;  // positions of the observer list
  const int a_lo = min(wv * per_wave, nW), a_hi = min(a_lo + per_wave, nW);
  unsigned* s_minb = s_minb_all + (size_t)wv * wave_words;  // [G * NL] minima | [G * V] candidates (u16) | [G * V] distance | [G * V] speed
  unsigned short* cand = reinterpret_cast<unsigned short*>(s_minb + (size_t)G * NL);
  float* rDist = reinterpret_cast<float*>(s_minb + (size_t)G * (NL + (V + 1) / 2));  // (the last two only with neighbour rows)
  float* rSpd = rDist + (size_t)G * V;
  int* pAO = reinterpret_cast<int*>(pDist_all[wv]);  // first beam << 16 | (observer of the round << 8) | body, per lane of the pass
  int* pPref = pPref_all[wv];
  for (int g0 = a_lo; g0 < a_hi; g0 += G) {
    const int g1 = min(g0 + G, a_hi);
    const int P = (g1 - g0) * nB;
    // (two words per lane and instruction where the area allows it: 240 beams x 8 observers are 30 single-word rounds)
    const bool two = (NL & 1) == 0 && (reinterpret_cast<uintptr_t>(s_minb) & 7u) == 0u;
    if (two) {
      const uint2 ones = make_uint2(__float_as_uint(1.0f), __float_as_uint(1.0f));
      for (int k = lane; k < (g1 - g0) * (NL >> 1); k += WAVE) reinterpret_cast<uint2*>(s_minb)[k] = ones;
    } else
    for (int k = lane; k < (g1 - g0) * NL; k += WAVE) s_minb[k] = __float_as_uint(1.0f);
    row_sync<true>();  // the round belongs to this wave alone
    PHASE_MARK(21);  // env obs: per-beam minima initialised
    // stage 1, all pairs of the round: which bodies are within the lidar's reach of which observer (the reference's broad phase);
    // the few that are go into a list, and only the list pays for beam windows, prefix sums and casting -- two thirds of the
    // pairs of a 30-agent roundabout fail this test, and a pass of 64 pairs cost the same whether one lane passed or all
    // (a round of at most 64 pairs -- 8 agents -- is its own list: the verdicts stay in the lanes)
    const bool direct = P <= WAVE;
    const float inv_nB = 1.0f / (float)(nB > 0 ? nB : 1);
    bool in_d = false;
    int ao_d = 0;
    int n_cand = 0;  // (uniform)
    for (int q0 = 0; q0 < P; q0 += WAVE) {
      const int pq = q0 + lane;
      const bool pv = pq < P;
      const int al = pv ? div_small(pq, inv_nB) : 0, bo = pv ? pq - al * nB : 0;
      const int a = wList[g0 + al], o = bList[bo];  // observer and body of the pair
      bool in = false;
      if (pv) {
        const int stt = bST[o] & 0xff, kind = bST[o] >> 8;
        bool present = stt == ST_PENDING || stt == ST_ACTIVE || stt == ST_DYING;
        bool still = stt == ST_DYING;  // a finished agent is a static body (zero velocity)
        if (flags && o < A) {
          // multi-agent step: rows of agents that drove this step show the world before the finishes / respawns
          // (base_env.py:303-344 runs before multi_agent_pgdrive.py:128-141); an agent spawned this step sees the world at
          // its spawn time, i.e. the earlier spawns of the step only
          const uint32_t fa = bFL[a], fo = bFL[o];
          if (fa & PGD_F_RESET) {
          } else if (fa & PGD_F_NEW) {
            present = present && (!(fo & PGD_F_NEW) || bAID[o] < bAID[a]);
          } else {
            present = (fo & PGD_F_REPORT) || (present && !(fo & PGD_F_NEW));
            still = still && !(fo & PGD_F_REPORT);
          }
        }
        const float px = bX[a], py = bY[a];
        const float x = bX[o], y = bY[o];
        in = present && o != a && shape_point_dist<OBJ>(Obb{x, y, bUX[o], bUY[o], bHL[o], bHW[o]}, px, py) <= R;
        if (NO > 0) {  // kept for the neighbour ranks of the round
          rDist[pq] = (in && kind == PGD_OBJ_VEHICLE) ? norm2(px - x, py - y) : __builtin_inff();
          rSpd[pq] = (in && !still) ? speed_kmh(bV[o]) : 0.0f;
        }
      }
      const unsigned long long im = __ballot(in);
      if (direct) {
        in_d = in; ao_d = (al << 8) | o;
        n_cand = im != 0ull ? 1 : 0;
      } else {
        if (in) cand[n_cand + __popcll(im & ((1ull << lane) - 1ull))] = (unsigned short)((al << 8) | o);
        n_cand += __popcll(im);
      }
    }
    if (!direct) row_sync<true>();
    PHASE_MARK(22);  // env obs: pairs (broad phase)
    // stage 2, the pairs within reach, 64 at a time: beam window of the body (the arithmetic of obs_compact), then its incidences
    for (int q0 = 0; q0 < n_cand; q0 += WAVE) {
      const bool pv = direct ? in_d : q0 + lane < n_cand;
      const int ao_c = direct ? ao_d : (pv ? (int)cand[q0 + lane] : 0);
      const int al = ao_c >> 8, o = ao_c & 0xff;
      const int a = wList[g0 + al];
      int i0 = 0, cnt = 0;
      if (pv) {
        const float px = bX[a], py = bY[a], hx = bUX[a], hy = bUY[a];
        const float x = bX[o], y = bY[o], hl = bHL[o], hw = bHW[o];
        const float dist = norm2(px - x, py - y);
        const float rad = (hw < 0.0f ? hl : norm2(hl, hw)) * 1.02f + 0.01f;
        i0 = 0; cnt = NL;
        if (dist > rad) {
          const float rx = (x - px) * hx + (y - py) * hy, ry = (y - py) * hx - (x - px) * hy;
          beam_window(rx, ry, rad / dist, NL, i0, cnt);
        }
      }
      int inc = cnt;  // inclusive prefix sum of the window sizes over the wave
#pragma unroll
      for (int sh = 1; sh < WAVE; sh <<= 1) {
        const int up = __shfl_up(inc, sh);
        if (lane >= sh) inc += up;
      }
      const int T = __shfl(inc, WAVE - 1);
      const int R = (T + WAVE - 1) / WAVE;  // rounds of 64 incidences
      // lidar (distance_detector.py:65-94, cutils.pyx:60-142): incidence t belongs to the pair whose window covers it
      auto cast = [&](const int ao, const int start, const int t) {
        const int i0w = ao >> 16, qa = (ao >> 8) & 0xff, qo = ao & 0xff, ga = wList[g0 + qa];
        int i = i0w + (t - start);
        i -= i >= NL ? NL : 0;
        const float ax = bX[ga], ay = bY[ga], ahx = bUX[ga], ahy = bUY[ga];
        const float2 bd = d.beam[i];  // (cos, sin)(i * 2 pi / NL); rotated by the heading
        const float dx = R_lidar * (bd.x * ahx - bd.y * ahy), dy = R_lidar * (bd.y * ahx + bd.x * ahy);
        const float f = shape_ray<OBJ>(Obb{bX[qo], bY[qo], bUX[qo], bUY[qo], bHL[qo], bHW[qo]}, ax, ay, dx, dy);
        atomicMin(&s_minb[qa * NL + i], __float_as_uint(f));
      };
      if (R <= 32) {
        // Owner of an incidence without a search: the non-empty pairs are compacted (window start, first beam, observer | body),
        // and a 64-bit mask per round marks the positions at which a window starts.  Incidence t of round r then belongs to
        // compacted pair (pairs that start before the round) + (starts at or before t's position in the round) - 1: one
        // uniform mask read and a population count instead of a six-step binary search through LDS per incidence.
        unsigned long long* pMask = M.pMask[wv];
        const unsigned long long nz = __ballot(cnt > 0);
        if (lane < R) pMask[lane] = 0ull;
        row_sync<true>();
        if (cnt > 0) {
          const int k = __popcll(nz & ((1ull << lane) - 1ull));
          const int start = inc - cnt;
          pAO[k] = (i0 << 16) | (al << 8) | o;
          pPref[k] = start;
          atomicOr(&pMask[start >> 6], 1ull << (start & 63));
        }
        row_sync<true>();
        int before = 0;  // non-empty pairs whose window starts before the round
        for (int r = 0; r < R; ++r) {
          const unsigned long long m = pMask[r];
          const int t = r * WAVE + lane;
          const int k = before + __popcll(m & ((2ull << lane) - 1ull)) - 1;
          before += __popcll(m);
          if (t < T) cast(pAO[k], pPref[k], t);
        }
      } else {
        pAO[lane] = (i0 << 16) | (al << 8) | o;
        pPref[lane + 1] = inc;
        if (lane == 0) pPref[0] = 0;
        row_sync<true>();
        for (int t = lane; t < T; t += WAVE) {  // pPref[p] <= t < pPref[p + 1]
          int pl = 0;
#pragma unroll
          for (int sh = WAVE / 2; sh > 0; sh >>= 1)
            if (pPref[pl + sh] <= t) pl += sh;
          cast(pAO[pl], pPref[pl], t);
        }
      }
      row_sync<true>();
    }
    PHASE_MARK(23);  // env obs: incidences cast
    // neighbour rows (lidar.py:55-77: by centre distance, stable in slot order); the multi-agent default observes none
    if (NO > 0)
      for (int pq = lane; pq < P; pq += WAVE) {
        const int al = div_small(pq, inv_nB), bo = pq - al * nB, a = wList[g0 + al], o = bList[bo];
        const float dk = rDist[pq];
        int rank = 0, nveh = 0;
        for (int j = 0; j < nB; ++j) {  // (the body list ascends: position order = slot order, the reference's tie rule)
          const float dj = rDist[al * nB + j];
          nveh += dj < __builtin_inff() ? 1 : 0;
          rank += (dj < dk || (dj == dk && j < bo)) ? 1 : 0;
        }
        float* nb = obs + (size_t)e * d.ostride + (size_t)a * D + o_oth;
        if (oth) {
          if (dk < __builtin_inff() && rank < NO) { nbSlot[a * NO + rank] = o; nbSpd[a * NO + rank] = rSpd[pq]; }
          continue;
        }
        if (dk < __builtin_inff() && rank < NO) {
          const float px = bX[a], py = bY[a], hx = bUX[a], hy = bUY[a], spd = rSpd[pq];
          const float ms = aMS[a], sp_me = speed_kmh(bV[a]);
          float ph, ps;
          projection(hx, hy, bX[o] - px, bY[o] - py, ph, ps);
          float* w = nb + rank * 4;
          w[0] = clipf((ph / R + 1.0f) * 0.5f, 0.0f, 1.0f);
          w[1] = clipf((ps / R + 1.0f) * 0.5f, 0.0f, 1.0f);
          projection(hx, hy, spd * bUX[o] - sp_me * hx, spd * bUY[o] - sp_me * hy, ph, ps);
          w[2] = clipf((ph / ms + 1.0f) * 0.5f, 0.0f, 1.0f);
          w[3] = clipf((ps / ms + 1.0f) * 0.5f, 0.0f, 1.0f);
        }
        for (int r = nveh + bo; r < NO; r += nB) {  // absent neighbours -> zeros
          float* w = nb + r * 4;
          w[0] = w[1] = w[2] = w[3] = 0.0f;
        }
      }
    // without lidar noise a fan is copied two beams per lane and instruction (the row's beams start at an even column of an even-length
    // row in every multi-agent layout that has this property: checked, else one by one): the write-out of 8 x 240 beams was 8.5 k cycles
    const bool plain = !(d.cfg.lidar_gaussian_noise > 0.0f || d.cfg.lidar_dropout_prob > 0.0f);
    const int o_lid = o_oth + (oth ? o_oth : 4) * NO;
    const bool pairs_ok = two && plain && ((D | o_lid | d.ostride) & 1) == 0 && (reinterpret_cast<uintptr_t>(obs) & 7u) == 0u;
    if (pairs_ok && NL >= 2 * WAVE) {
      for (int qa = 0; qa < g1 - g0; ++qa) {
        const int ga = wList[g0 + qa];
        obs_f2* dst = reinterpret_cast<obs_f2*>(obs + (size_t)e * d.ostride + (size_t)ga * D + o_lid);
        const obs_f2* src = reinterpret_cast<const obs_f2*>(s_minb + qa * NL);
        for (int i = lane; i < (NL >> 1); i += WAVE) OBS_ST_FAN(dst + i, src[i]);
      }
    } else if (pairs_ok) {
      const int half = NL >> 1;
      const float inv_half = 1.0f / (float)half;
      for (int k = lane; k < (g1 - g0) * half; k += WAVE) {
        const int qa = div_small(k, inv_half), i = k - qa * half, ga = wList[g0 + qa];
        OBS_ST_FAN(reinterpret_cast<obs_f2*>(obs + (size_t)e * d.ostride + (size_t)ga * D + o_lid) + i, reinterpret_cast<const obs_f2*>(s_minb)[k]);
      }
    } else
    if (NL >= 2 * WAVE) {  // long fans observer by observer: no division per element (240 beams: 116 -> 123 M env-steps/s at 8 agents)
      for (int qa = 0; qa < g1 - g0; ++qa) {
        const int ga = wList[g0 + qa];
        float* dst = obs + (size_t)e * d.ostride + (size_t)ga * D + o_oth + (oth ? o_oth : 4) * NO;
        for (int i = lane; i < NL; i += WAVE) dst[i] = lidar_noise(d, e, ga, tick, i, __uint_as_float(s_minb[qa * NL + i]));
      }
    } else  // short fans flat over (observer, beam): a round per observer would leave most lanes of its second round idle
    for (int k = lane; k < (g1 - g0) * NL; k += WAVE) {
      const int qa = div_small(k, 1.0f / (float)NL), i = k - qa * NL, ga = wList[g0 + qa];
      obs[(size_t)e * d.ostride + (size_t)ga * D + o_oth + (oth ? o_oth : 4) * NO + i] = lidar_noise(d, e, ga, tick, i, __uint_as_float(s_minb[k]));
    }
    row_sync<true>();
    PHASE_MARK(24);  // env obs: neighbour rows, lidar rows written
  }
  if (oth) {  // the neighbours' state vectors, every observer at once (the lane groups of the state phase)
    __syncthreads();
    if (want)
      for (int rk = 0; rk < NO; ++rk) {
        const int o = nbSlot[sa * NO + rk];
        float* dst = row + o_oth + rk * o_oth;
        if (o < 0) {
          for (int k = st; k < o_oth; k += LPA) dst[k] = 0.0f;
        } else {
          const AgentView oa = view_of_slot(d, mv, recs, spb, o, nbSpd[sa * NO + rk], e, tick);
          state_block<false>(d, mv, spb[rec_spawn(recs, d.V, o)], oa, dst, st, LPA);
        }
      }
  }
}


#endif
