// pgd_engine.hip — kernels + C ABI (include/pgdrive_hip.h) of the MI355X-native batched PGDrive step engine.
//
// Execution model (gfx950, wave64):
//   k_step     one 64-lane wave per block; lane = one vehicle slot, a wave carries floor(64/V) whole environments.
//              IDM neighbour search, crash test and trigger logic read the env's vehicle snapshot from LDS;
//              localisation / line / sidewalk tests walk the per-map uniform grid (L2-resident, immutable).
//              Reward, done and auto-reset are fused at the end, so one launch advances every vehicle 0.1 s.
//   k_observe  one 256-thread block per (env, agent): vehicle boxes of the env are compacted into LDS with a wave
//              ballot, one lidar beam per thread (min over LDS-broadcast boxes), 274-float row written coalesced.
// The reference call stack this replaces: envs/base_env.py:184-224,303-344.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "pgd_device.h"

#define HIPCHK(x)                                                                              \
  do {                                                                                         \
    hipError_t _e = (x);                                                                       \
    if (_e != hipSuccess) {                                                                    \
      fprintf(stderr, "[pgdrive_hip] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
      return PGD_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

#define WAVE 64
#define MAXV 64

// Optional per-phase cycle counters of k_step (build with -DPGD_PROF; never enabled in the shipped library)
#ifdef PGD_PROF
#define PROF_BLOCKS 8192
__device__ unsigned long long g_phase_cycles[PROF_BLOCKS * 32];  // per block, no atomics (they would serialise)
__shared__ long long s_prof_t0, s_prof_w0;
#define PHASE_MARK(k)                                                                                      \
  do {                                                                                                     \
    if ((int)threadIdx.x == __builtin_ffsll((long long)__ballot(1)) - 1 && blockIdx.x < PROF_BLOCKS) {       \
      long long _now = clock64();                                                                          \
      g_phase_cycles[blockIdx.x * 32 + (k)] += (unsigned long long)(_now - s_prof_t0);                     \
      s_prof_t0 = _now;                                                                                    \
    }                                                                                                      \
  } while (0)
#define PHASE_INIT() do { if (threadIdx.x == 0) { s_prof_t0 = clock64(); s_prof_w0 = wall_clock64(); } } while (0)
#define PHASE_END() do { if (threadIdx.x == 0 && blockIdx.x < PROF_BLOCKS) g_phase_cycles[blockIdx.x * 32 + 15] += (unsigned long long)(wall_clock64() - s_prof_w0); } while (0)
#else
#define PHASE_MARK(k)
#define PHASE_INIT()
#define PHASE_END()
#endif

// ---------------------------------------------------------------------------------------------------------------------
// per-lane vehicle registers
// ---------------------------------------------------------------------------------------------------------------------
struct Veh {
  float x, y, th, v, steer, thr, lastx, lasty, lasthx, lasthy, a0s, a0t, a1s, a1t, php, phi, plp, pli, target, energy,
      dl, dr, eprew;
  int status, lane, ck0, ck1, rlane, timer, vflags, spawn;
  float agent_id;
  float hx, hy;  // unit heading (cos, sin of th): derived, kept in registers, never stored
};

DEV void load_veh(const PgdDev& d, int e, int s, Veh& r) {
  VehRec t;
  const uint4* src = reinterpret_cast<const uint4*>(d.rec + (size_t)e * d.V + s);
  uint4* dst = reinterpret_cast<uint4*>(&t);
#pragma unroll
  for (int k = 0; k < 8; ++k) dst[k] = src[k];
  r.x = t.f[SF_X]; r.y = t.f[SF_Y]; r.th = t.f[SF_THETA]; r.v = t.f[SF_SPEED];
  r.steer = t.f[SF_STEER]; r.thr = t.f[SF_THROTTLE];
  r.lastx = t.f[SF_LASTX]; r.lasty = t.f[SF_LASTY]; r.lasthx = t.f[SF_LASTHX]; r.lasthy = t.f[SF_LASTHY];
  r.a0s = t.f[SF_ACT0S]; r.a0t = t.f[SF_ACT0T]; r.a1s = t.f[SF_ACT1S]; r.a1t = t.f[SF_ACT1T];
  r.php = t.f[SF_PID_HP]; r.phi = t.f[SF_PID_HI]; r.plp = t.f[SF_PID_LP]; r.pli = t.f[SF_PID_LI];
  r.target = t.f[SF_TARGET_SPEED]; r.energy = t.f[SF_ENERGY];
  r.dl = t.f[SF_DIST_LEFT]; r.dr = t.f[SF_DIST_RIGHT]; r.eprew = t.f[SF_EP_REWARD];
  r.status = t.i[SI_STATUS]; r.lane = t.i[SI_LANE]; r.ck0 = t.i[SI_CK0]; r.ck1 = t.i[SI_CK1];
  r.rlane = t.i[SI_RLANE]; r.timer = t.i[SI_TIMER]; r.vflags = t.i[SI_VFLAGS]; r.spawn = t.i[SI_SPAWN];
  r.agent_id = t.f[SF_AGENT_ID];
  sincosf(r.th, &r.hy, &r.hx);
}
DEV void store_veh(const PgdDev& d, int e, int s, const Veh& r) {
  VehRec t;
  t.f[SF_X] = r.x; t.f[SF_Y] = r.y; t.f[SF_THETA] = r.th; t.f[SF_SPEED] = r.v;
  t.f[SF_STEER] = r.steer; t.f[SF_THROTTLE] = r.thr;
  t.f[SF_LASTX] = r.lastx; t.f[SF_LASTY] = r.lasty; t.f[SF_LASTHX] = r.lasthx; t.f[SF_LASTHY] = r.lasthy;
  t.f[SF_ACT0S] = r.a0s; t.f[SF_ACT0T] = r.a0t; t.f[SF_ACT1S] = r.a1s; t.f[SF_ACT1T] = r.a1t;
  t.f[SF_PID_HP] = r.php; t.f[SF_PID_HI] = r.phi; t.f[SF_PID_LP] = r.plp; t.f[SF_PID_LI] = r.pli;
  t.f[SF_TARGET_SPEED] = r.target; t.f[SF_ENERGY] = r.energy;
  t.f[SF_DIST_LEFT] = r.dl; t.f[SF_DIST_RIGHT] = r.dr; t.f[SF_EP_REWARD] = r.eprew; t.f[SF_AGENT_ID] = r.agent_id;
  t.i[SI_STATUS] = r.status; t.i[SI_LANE] = r.lane; t.i[SI_CK0] = r.ck0; t.i[SI_CK1] = r.ck1;
  t.i[SI_RLANE] = r.rlane; t.i[SI_TIMER] = r.timer; t.i[SI_VFLAGS] = r.vflags; t.i[SI_SPAWN] = r.spawn;
  uint4* dst = reinterpret_cast<uint4*>(d.rec + (size_t)e * d.V + s);
  const uint4* src = reinterpret_cast<const uint4*>(&t);
#pragma unroll
  for (int k = 0; k < 8; ++k) dst[k] = src[k];
}

// base_vehicle.py:394-401; the magnitude: a reversing vehicle has a negative speed field, and BaseVehicle.velocity is this
// magnitude times the FORWARD vector even then (base_vehicle.py:419-425)
DEV float speed_kmh(float v) { return clipf(fabsf(v) * 3.6f, 0.0f, 100000.0f); }

// env snapshot in LDS (one entry per lane of the wave)
struct Snap {
  float x[WAVE], y[WAVE], ux[WAVE], uy[WAVE], spd[WAVE], hl[WAVE], hw[WAVE];
  int lane[WAVE], present[WAVE];
  // for the IDM neighbour search: each vehicle's longitudinal coordinate on its own lane, that lane's length and
  // successor list (8 x int16), so the O(V^2) search never touches the lane table
  float lon[WAVE], llen[WAVE];
  int4 succ[WAVE];
};
DEV bool succ_has(const int4& p, int id) {  // 8 packed int16 ids, unused entries are -1
  unsigned u = (unsigned)id & 0xffffu;
  unsigned a = (unsigned)p.x, b = (unsigned)p.y, c = (unsigned)p.z, d = (unsigned)p.w;
  return (a & 0xffffu) == u || (a >> 16) == u || (b & 0xffffu) == u || (b >> 16) == u || (c & 0xffffu) == u ||
         (c >> 16) == u || (d & 0xffffu) == u || (d >> 16) == u;
}
DEV Obb snap_obb(const Snap& S, int k) { return Obb{S.x[k], S.y[k], S.ux[k], S.uy[k], S.hl[k], S.hw[k]}; }

// ---------------------------------------------------------------------------------------------------------------------
// sub-lane cooperation: a vehicle is carried by SUB consecutive lanes that hold identical copies of its registers; the
// heavy box / neighbour loops are split across them and recombined with wave shuffles (all lanes of a group are always
// convergent because they execute on identical data).
// ---------------------------------------------------------------------------------------------------------------------
struct Grp {
  int sub, SUB, lead;
};
DEV unsigned group_min(unsigned v, const Grp& g) {
  unsigned r = v;
  for (int j = 0; j < g.SUB; ++j) r = min(r, (unsigned)__shfl((int)v, g.lead + j));
  return r;
}
DEV unsigned group_or(unsigned v, const Grp& g) {
  unsigned r = v;
  for (int j = 0; j < g.SUB; ++j) r |= (unsigned)__shfl((int)v, g.lead + j);
  return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// localisation: utils/scene_utils.py:138-185 + navigation.py:328-344.  "First hit" = smallest box id (Bullet insertion
// order); the cell-major box copies keep that order, so the smallest list position per class is the answer.
// key = (position in cell << 16) | lane id
// ---------------------------------------------------------------------------------------------------------------------
// Device-side cell index (built by pgd_upload_maps): inside a cell the lane-surface boxes come first (original relative
// order), the line / sidewalk boxes follow.  cstart[c] = first item | (number of lane boxes << 24); the cell ends where the
// next one starts.  Localisation scans only the lane part, the contact / ray tests only the rest.
DEV float ray_grid(const MapView& mv, float px, float py, float dx, float dy, unsigned kinds);
DEV int cell_first(int c) { return c & 0xffffff; }
DEV int cell_mid(int c) { return (c & 0xffffff) + (int)((unsigned)c >> 24); }

DEV int get_current_lane(const MapView& mv, const Grp& g, float px, float py, float hx, float hy, int road_cur,
                         int road_next) {
  const pgd_map& m = *mv.m;
  int cx = (int)floorf((px - m.ox) / m.cell), cy = (int)floorf((py - m.oy) / m.cell);
  int k0 = 0, k1 = 0;
  if (cx >= 0 && cy >= 0 && cx < m.gx && cy < m.gy) {
    const int c = mv.cstart[cy * m.gx + cx];
    k0 = cell_first(c);
    k1 = cell_mid(c);
  }
  unsigned best_cur = 0xffffffffu, best_next = 0xffffffffu, best_any = 0xffffffffu;
  const int stride = g.SUB;
  constexpr int NB = 3;  // boxes per sub-lane and round: a cell holds 3-9 lane boxes, so one round is the rule
  for (int k = k0 + g.sub; k < k1; k += NB * stride) {
    pgd_box b[NB];
    LaneExt x[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      int kk = k + j * stride;
      kk = kk < k1 ? kk : k;
      b[j] = mv.cbox[kk];  // batch the independent loads
      x[j] = mv.cext[kk];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      int kk = k + j * stride;
      if (kk >= k1) continue;
      if (!point_in_obb(obb_of(b[j]), px, py)) continue;
      unsigned key = ((unsigned)(kk - k0) << 16) | (unsigned)b[j].lane;
      bool is_cur = x[j].road == road_cur, is_next = x[j].road == road_next;
      if (!(key < best_any || (is_cur && key < best_cur) || (is_next && key < best_next))) continue;
      // cos(angle between lane heading at the point and vehicle heading) > 0 (scene_utils.py:158-172); only the sign is
      // used, so the lane direction is taken in closed form: straight = unit dir; arc = dir * (-dy, dx) around the centre
      float dirx, diry;
      if (x[j].dir == 0.0f) { dirx = x[j].ax; diry = x[j].ay; }
      else { dirx = -x[j].dir * (py - x[j].ay); diry = x[j].dir * (px - x[j].ax); }
      if (!(dirx * hx + diry * hy > 0.0f)) continue;
      best_any = min(best_any, key);
      if (is_cur) best_cur = min(best_cur, key);
      if (is_next) best_next = min(best_next, key);
    }
  }
  best_cur = group_min(best_cur, g);
  best_next = group_min(best_next, g);
  best_any = group_min(best_any, g);
  unsigned pick = best_cur != 0xffffffffu ? best_cur : (road_next < 0 ? best_any : (best_next != 0xffffffffu ? best_next : best_any));
  return pick == 0xffffffffu ? -1 : (int)(pick & 0xffffu);
}

// Navigation._update_target_checkpoints (navigation.py:262-282)
DEV void update_checkpoints(const MapView& mv, const pgd_spawn& sp, Veh& r, float lon) {
  if (r.ck0 == r.ck1) return;
  if (!(lon < 5.0f)) return;
  int n = sp.n_ckpt;
  int start_node = mv.roads[mv.lanes[r.lane].road].from;
  bool in_tail = false;
  int idx = -1;
  for (int k = r.ck1; k < n; ++k) {
    if (sp.ckpt[k] == start_node) {
      in_tail = true;
      if (idx < 0 && k < n - 1) idx = k;
    }
  }
  if (!in_tail || idx < 0) return;
  r.ck0 = idx;
  r.ck1 = (idx + 1 == n - 1) ? idx : idx + 1;
}

// Navigation.update_localization (navigation.py:155-183)
DEV void update_localization(const MapView& mv, const Grp& g, const pgd_spawn& sp, Veh& r) {
  const float s = r.hy, c = r.hx;
  int road_cur = sp.ckpt_road[r.ck0];
  int road_next = (r.ck0 == r.ck1) ? -1 : sp.ckpt_road[r.ck1];
  PHASE_MARK(16);  // after_step: route roads
  int lane = get_current_lane(mv, g, r.x, r.y, c, s, road_cur, road_next);
  PHASE_MARK(17);  // after_step: get_current_lane
  bool on_lane = lane >= 0;
  if (!on_lane) lane = r.lane;
  r.lane = lane;
  float lon, lat;
  lane_local(mv.lanes[lane], r.x, r.y, lon, lat);
  update_checkpoints(mv, sp, r, lon);
  r.vflags = on_lane ? (r.vflags & ~PGD_F_OFF_LANE) : (r.vflags | PGD_F_OFF_LANE);
  PHASE_MARK(18);  // after_step: lane_local + checkpoints
}

// BaseVehicle._state_check (base_vehicle.py:615-644)
DEV unsigned state_check(const MapView& mv, const Grp& g, const Obb& car) {
  const pgd_map& m = *mv.m;
  float ex = fabsf(car.ux) * car.hl + fabsf(car.uy) * car.hw, ey = fabsf(car.uy) * car.hl + fabsf(car.ux) * car.hw;
  int cx0 = max((int)floorf((car.cx - ex - m.ox) / m.cell), 0), cx1 = min((int)floorf((car.cx + ex - m.ox) / m.cell), m.gx - 1);
  int cy0 = max((int)floorf((car.cy - ey - m.oy) / m.cell), 0), cy1 = min((int)floorf((car.cy + ey - m.oy) / m.cell), m.gy - 1);
  unsigned fl = 0;
  const int stride = g.SUB;
  for (int cy = cy0; cy <= cy1; ++cy)
    for (int cx = cx0; cx <= cx1; ++cx) {
      int cell = cy * m.gx + cx;
      int k0 = cell_mid(mv.cstart[cell]), k1 = cell_first(mv.cstart[cell + 1]);
      for (int k = k0 + g.sub; k < k1; k += 4 * stride) {
        pgd_box b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int kk = k + j * stride;
          b[j] = mv.cbox[kk < k1 ? kk : k];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int kk = k + j * stride;
          if (kk >= k1) continue;
          unsigned bit = b[j].kind == PGD_BOX_WHITE ? PGD_F_ON_WHITE
                         : b[j].kind == PGD_BOX_YELLOW ? PGD_F_ON_YELLOW
                         : b[j].kind == PGD_BOX_BROKEN ? PGD_F_ON_BROKEN : PGD_F_CRASH_SIDEWALK;
          if (fl & bit) continue;
          if (obb_overlap(car, obb_of(b[j]))) fl |= bit;
        }
      }
    }
  return group_or(fl, g);
}

// What the later phases need from the agent's route position (Navigation.current_ref_lanes / next_ref_lanes,
// navigation.py:155-183): looked up once per step after the checkpoint update, then reused by the side distances, the
// reward and the observation instead of re-walking spawn record -> road table each time.
struct RouteCtx {
  int blk;         // Road.block_ID char of the current road
  int road_cur;    // road id of checkpoints[ck0] -> checkpoints[ck0 + 1]
  int cur_first;   // its first lane (current_ref_lanes[0]) ...
  int cur_n;       // ... and lane count
  int next_first;  // first lane of the next checkpoint road (== cur_first on the last road)
};
DEV RouteCtx route_ctx(const MapView& mv, const pgd_spawn& sp, int ck0, int ck1) {
  const int rc = sp.ckpt_road[ck0], rn = sp.ckpt_road[ck1];
  const pgd_road& CR = mv.roads[rc];
  const pgd_road& NR = mv.roads[rn];
  return RouteCtx{CR.block_id, rc, CR.first_lane, CR.n_lanes, NR.first_lane};
}

// BaseVehicle.after_step (base_vehicle.py:255-290).  `with_state_check` = false lets the caller run the line / sidewalk
// test wave-cooperatively afterwards (k_step with one env per wave) and OR the result into vflags.
DEV void after_step_vehicle(const MapView& mv, const Grp& g, const pgd_spawn& sp, Veh& r, bool is_agent,
                            bool with_state_check, RouteCtx& ctx) {
  update_localization(mv, g, sp, r);
  if (is_agent) {
    ctx = route_ctx(mv, sp, r.ck0, r.ck1);
    unsigned fl = (unsigned)r.vflags;
    fl &= ~(PGD_F_ON_WHITE | PGD_F_ON_YELLOW | PGD_F_ON_BROKEN | PGD_F_CRASH_SIDEWALK | PGD_F_OUT_OF_ROUTE);
    if (with_state_check) fl |= state_check(mv, g, Obb{r.x, r.y, r.hx, r.hy, 0.5f * sp.length, 0.5f * sp.width});
    float lon, lat;
    const pgd_lane& L0 = mv.lanes[ctx.cur_first];
    lane_local(L0, r.x, r.y, lon, lat);
    float w = mv.m->lane_width;
    r.dl = lat + w * 0.5f;
    float range = w * ctx.cur_n;
    if (ctx.blk == 'y' || ctx.blk == 'Y') {
      // Navigation.get_current_lateral_range on Merge / Split blocks (navigation.py:306-320,346-362): a 50 m ray from the
      // left edge of the leftmost reference lane across the road against the continuous lane lines
      float sx, sy;
      lane_position(L0, lon, -0.5f * L0.width, sx, sy);
      range = 50.0f * ray_grid(mv, sx, sy, -L0.by * 50.0f, L0.bx * 50.0f, (1u << PGD_BOX_WHITE) | (1u << PGD_BOX_YELLOW));
    }
    r.dr = range - r.dl;
    if (r.dr < 0.0f || r.dl < 0.0f) fl |= PGD_F_OUT_OF_ROUTE;
    r.vflags = (int)fl;
    float dist = norm2(r.lastx - r.x, r.lasty - r.y) / 1000.0f;
    r.energy += 3.25f * expf(0.01f * speed_kmh(r.v)) * dist / 100.0f * 1000.0f;
    PHASE_MARK(19);  // after_step: side distances
  }
}

// the same test with the whole wave on one car: the (<= 2x2) grid cells under the car are flattened into one index range
DEV unsigned state_check_wave(const MapView& mv, const Obb& car) {
  const pgd_map& m = *mv.m;
  const int lane = threadIdx.x;
  float ex = fabsf(car.ux) * car.hl + fabsf(car.uy) * car.hw, ey = fabsf(car.uy) * car.hl + fabsf(car.ux) * car.hw;
  int cx0 = max((int)floorf((car.cx - ex - m.ox) / m.cell), 0), cx1 = min((int)floorf((car.cx + ex - m.ox) / m.cell), m.gx - 1);
  int cy0 = max((int)floorf((car.cy - ey - m.oy) / m.cell), 0), cy1 = min((int)floorf((car.cy + ey - m.oy) / m.cell), m.gy - 1);
  unsigned fl = 0;
  for (int cyb = cy0; cyb <= cy1; cyb += 2)
    for (int cxb = cx0; cxb <= cx1; cxb += 2) {  // blocks of up to 2x2 cells (a car spans at most 2 cells per axis)
      int k0[4], pre[5];
      pre[0] = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int cx = cxb + (q & 1), cy = cyb + (q >> 1);
        bool in = cx <= cx1 && cy <= cy1;
        int cell = in ? cy * m.gx + cx : 0;
        int a = cell_mid(mv.cstart[cell]), b = cell_first(mv.cstart[cell + 1]);
        k0[q] = a;
        pre[q + 1] = pre[q] + (in ? b - a : 0);
      }
      for (int f = lane; f < pre[4]; f += WAVE) {
        int q = (f >= pre[1]) + (f >= pre[2]) + (f >= pre[3]);
        int kk = (q == 0 ? k0[0] : q == 1 ? k0[1] : q == 2 ? k0[2] : k0[3]) + f - (q == 0 ? pre[0] : q == 1 ? pre[1] : q == 2 ? pre[2] : pre[3]);
        pgd_box b = mv.cbox[kk];
        unsigned bit = b.kind == PGD_BOX_WHITE ? PGD_F_ON_WHITE
                       : b.kind == PGD_BOX_YELLOW ? PGD_F_ON_YELLOW
                       : b.kind == PGD_BOX_BROKEN ? PGD_F_ON_BROKEN : PGD_F_CRASH_SIDEWALK;
        if (obb_overlap(car, obb_of(b))) fl |= bit;
      }
    }
  unsigned out = 0;
  if (__ballot((fl & PGD_F_ON_WHITE) != 0)) out |= PGD_F_ON_WHITE;
  if (__ballot((fl & PGD_F_ON_YELLOW) != 0)) out |= PGD_F_ON_YELLOW;
  if (__ballot((fl & PGD_F_ON_BROKEN) != 0)) out |= PGD_F_ON_BROKEN;
  if (__ballot((fl & PGD_F_CRASH_SIDEWALK) != 0)) out |= PGD_F_CRASH_SIDEWALK;
  return out;
}

// ---------------------------------------------------------------------------------------------------------------------
// IDM: policy/idm_policy.py:82-133 (FrontBackObjects), :190-353 (act, lane change), :244-271 (PID + IDM law)
// ---------------------------------------------------------------------------------------------------------------------
struct Fbo {
  int front[3], back[3];
  float fd[3], bd[3];
  bool exist[3];
};

DEV void find_front_back(const MapView& mv, const Grp& g, const Snap& S, int base, int V, int self, unsigned long long objs,
                         int lane, float max_dist, bool with_ref, Fbo& r) {
  const pgd_lane& L = mv.lanes[lane];
  const int idx = L.index;  // the lanes of a road are consecutive; the device copy of the lane carries its road's lane count
  const int l0 = (with_ref && idx > 0) ? lane - 1 : -1;
  const int l2 = (with_ref && idx + 1 < L.pad) ? lane + 1 : -1;
  const float px = S.x[base + self], py = S.y[base + self];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    r.front[i] = r.back[i] = -1;
    r.fd[i] = r.bd[i] = max_dist;
  }
  r.exist[0] = l0 >= 0; r.exist[1] = true; r.exist[2] = l2 >= 0;
  // target lane t (0 left, 1 own, 2 right) is searched by sub-lane t mod SUB; t is a per-lane runtime value so the
  // sub-lanes run the same instructions on different target lanes (no serialisation across targets)
  for (int t = g.sub; t < 3; t += g.SUB) {
    const int tl = t == 0 ? l0 : (t == 1 ? lane : l2);
    if (tl < 0) continue;
    const pgd_lane& li = mv.lanes[tl];
    float cur, lat;
    lane_local(li, px, py, cur, lat);
    const float left_long = li.length - cur;
    const int4 lsucc = *reinterpret_cast<const int4*>(li.succ);
    // one pass, five running minima (FrontBackObjects.get_find_front_back_objs, idm_policy.py:107-131):
    //   same lane front/back; successor-lane front; predecessor-lane back (all / excluding successor-lane objects)
    float same_f = max_dist, same_b = max_dist, succ_f = max_dist, pred_b = max_dist, pred_bx = max_dist;
    int o_same_f = -1, o_same_b = -1, o_succ_f = -1, o_pred_b = -1, o_pred_bx = -1;
    bool found_f = false, found_b = false;
    // only the vehicles inside the broad phase, in slot order (ties keep the first one like the reference's loop)
    for (unsigned long long m = objs; m != 0ull; m &= m - 1ull) {
      const int o = __builtin_ctzll(m);
      const int ol = S.lane[base + o];
      const float olon = S.lon[base + o], ollen = S.llen[base + o];
      const int4 osucc = S.succ[base + o];
      if (ol == tl) {
        float lg = olon - cur;
        if (same_f > lg && lg > 0.0f) { same_f = lg; o_same_f = o; found_f = true; }
        if (lg < 0.0f && fabsf(lg) < same_b) { same_b = fabsf(lg); o_same_b = o; found_b = true; }
      } else {
        const bool is_succ = succ_has(lsucc, ol);
        if (is_succ) {
          float lg = olon + left_long;
          if (succ_f > lg && lg > 0.0f) { succ_f = lg; o_succ_f = o; }
        }
        if (succ_has(osucc, tl)) {
          float lg = ollen - olon + cur;
          if (pred_b > lg) { pred_b = lg; o_pred_b = o; }
          if (!is_succ && pred_bx > lg) { pred_bx = lg; o_pred_bx = o; }
        }
      }
    }
    // objects on the lane itself take precedence; an object on a successor lane is only a "front" candidate while no
    // same-lane front object exists, and only then is it barred from being a "back" candidate (the reference's elif)
    const float fd = found_f ? same_f : succ_f;
    const int fo = found_f ? o_same_f : o_succ_f;
    const float bd = found_b ? same_b : (found_f ? pred_b : pred_bx);
    const int bo = found_b ? o_same_b : (found_f ? o_pred_b : o_pred_bx);
    if (t == 0) { r.fd[0] = fd; r.front[0] = fo; r.bd[0] = bd; r.back[0] = bo; }
    else if (t == 1) { r.fd[1] = fd; r.front[1] = fo; r.bd[1] = bd; r.back[1] = bo; }
    else { r.fd[2] = fd; r.front[2] = fo; r.bd[2] = bd; r.back[2] = bo; }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {  // every sub-lane gets every target lane's result
    int src = g.lead + (i % g.SUB);
    r.front[i] = __shfl(r.front[i], src);
    r.back[i] = __shfl(r.back[i], src);
    r.fd[i] = __shfl(r.fd[i], src);
    r.bd[i] = __shfl(r.bd[i], src);
  }
}

template <bool OBJ>
DEV void idm_act(const PgdDev& d, const MapView& mv, const Grp& g, const pgd_spawn& sp, const Snap& S, int base, int V,
                 int s, int e, uint32_t step_count, unsigned long long pmask, Veh& r, float& out_steer, float& out_acc) {
  const float NORMAL = 30.0f, CREEP = 5.0f, SAFE = 15.0f, MAXD = 30.0f;
  const int vlane = r.lane;
  int rt = r.rlane;
  // the three reads the routing decision needs are independent of each other: issue them together
  const int cur_road = sp.ckpt_road[r.ck0];
  const int vl_road = mv.lanes[vlane].road;
  const int rt_road = rt < 0 ? vl_road : (int)mv.lanes[rt].road;
  const pgd_road& CR = mv.roads[cur_road];
  bool success;
  // move_to_next_road (idm_policy.py:222-242)
  if (rt < 0) {
    rt = vlane;
    success = vl_road == cur_road;
  } else if (rt_road != cur_road) {
    success = false;
    const pgd_lane& RT = mv.lanes[rt];
    for (int k = 0; k < CR.n_lanes; ++k)
      if (lane_is_prev_of(RT, CR.first_lane + k)) { rt = CR.first_lane + k; success = true; break; }
  } else if (vl_road == cur_road && rt != vlane) {
    rt = vlane;
    r.timer = (int)(pgd_rng(d.cfg.seed, (uint32_t)e, (uint32_t)s, step_count) % 25u);
    success = true;
  } else success = true;
  // is the (new) routing lane on the current road?  first case: the vehicle lane's road; second: only a found lane of
  // CR; third and fourth: established by the branch conditions
  const bool in_cur = r.rlane < 0 ? (vl_road == cur_road) : (rt_road != cur_road ? success : true);
  r.rlane = rt;

  // Lidar.get_surrounding_objects (lidar.py:109-124)
  float px = S.x[base + s], py = S.y[base + s];
  unsigned long long objs = 0ull;
  // pmask = slots whose vehicle is in the physics world (wave-uniform with one env per wave: a scalar loop)
  for (unsigned long long m = pmask & ~(1ull << s); m != 0ull; m &= m - 1ull) {
    const int o = __builtin_ctzll(m);
    const Obb ob = snap_obb(S, base + o);
    const bool in = S.present[base + o] && shape_point_dist<OBJ>(ob, px, py) <= 50.0f;
    objs |= in ? (1ull << o) : 0ull;
  }

  PHASE_MARK(9);  // idm: routing + broad phase
  int front_obj = -1;
  float front_dist = 5.0f;
  int steer_lane = rt;
  float speed = S.spd[base + s];
  // one neighbour search for both branches of IDMPolicy.act (idm_policy.py:195-208): with the reference lanes when the
  // routing lane is on the current road, on the routing lane alone otherwise; the reference's failed assert (routing lane
  // not in ref lanes although move_to_next_road succeeded) falls back to "no front object, distance 5"
  const bool search = !success || in_cur;
  Fbo fb;
  if (search) find_front_back(mv, g, S, base, V, s, objs, rt, MAXD, success, fb);
  PHASE_MARK(10);  // idm: front/back search
  if (success && in_cur) {
    int idx = mv.lanes[rt].index;
    int n_cur = CR.n_lanes;
    int avail_lo = 0, avail_hi = n_cur - 1;
    bool decided = false;
    if (r.ck0 != r.ck1) {
      const pgd_road& NR = mv.roads[sp.ckpt_road[r.ck1]];
      int diff = n_cur - NR.n_lanes;
      if (diff > 0) {
        if (lane_is_prev_of(mv.lanes[CR.first_lane], NR.first_lane)) { avail_lo = 0; avail_hi = NR.n_lanes - 1; }
        else { avail_lo = diff; avail_hi = n_cur - 1; }
        if (idx < avail_lo || idx > avail_hi) {
          int side = idx > avail_hi ? 0 : 2;  // 0: change to left, 2: change to right
          // static indices only: a runtime index would push the whole Fbo into scratch memory
          const float side_bd = side == 0 ? fb.bd[0] : fb.bd[2], side_fd = side == 0 ? fb.fd[0] : fb.fd[2];
          const int side_front = side == 0 ? fb.front[0] : fb.front[2];
          if (side_bd < SAFE || side_fd < 5.0f) {
            r.target = CREEP;
            front_obj = fb.front[1]; front_dist = fb.fd[1]; steer_lane = rt;
          } else {
            r.target = NORMAL;
            front_obj = side_front; front_dist = side_fd;
            steer_lane = CR.first_lane + idx + (side == 0 ? -1 : 1);
          }
          decided = true;
        }
      }
    }
    if (!decided) {
      if (fabsf(speed - NORMAL) > 3.0f && fb.front[1] >= 0 && fabsf(S.spd[base + fb.front[1]] - NORMAL) > 3.0f &&
          r.timer > 50) {
        float fs = S.spd[base + fb.front[1]];
        bool has_r = false, has_l = false;
        float rs = 0.0f, ls = 0.0f;
        if (fb.front[2] >= 0) { has_r = true; rs = S.spd[base + fb.front[2]]; }
        else if (fb.exist[2] && fb.fd[2] > SAFE && fb.bd[2] > SAFE) { has_r = true; rs = 100.0f; }
        if (fb.front[0] >= 0) { has_l = true; ls = S.spd[base + fb.front[0]]; }
        else if (fb.exist[0] && fb.fd[0] > SAFE && fb.bd[0] > SAFE) { has_l = true; ls = 100.0f; }
        if (has_l && ls - fs > 10.0f) {
          int ex = idx - 1;
          if (ex >= avail_lo && ex <= avail_hi) {
            front_obj = fb.front[0]; front_dist = fb.fd[0]; steer_lane = CR.first_lane + ex; decided = true;
          }
        }
        if (!decided && has_r && rs - fs > 10.0f) {
          int ex = idx + 1;
          if (ex >= avail_lo && ex <= avail_hi) {
            front_obj = fb.front[2]; front_dist = fb.fd[2]; steer_lane = CR.first_lane + ex; decided = true;
          }
        }
      }
      if (!decided) {
        r.target = NORMAL;
        r.timer += 1;
        front_obj = fb.front[1]; front_dist = fb.fd[1]; steer_lane = rt;
      }
    }
  } else if (!success) {
    front_obj = fb.front[1]; front_dist = fb.fd[1]; steer_lane = rt;
  }
  PHASE_MARK(11);  // idm: lane-change logic

  // steering_control (idm_policy.py:244-252)
  const pgd_lane& SL = mv.lanes[steer_lane];
  float lon, lat;
  lane_local(SL, px, py, lon, lat);
  float lane_heading = lane_heading_at(SL, lon + 1.0f);
  float steering = pid_update(r.php, r.phi, 1.7f, 0.01f, 3.5f, wrap_to_pi(lane_heading - r.th));
  steering += pid_update(r.plp, r.pli, 0.3f, 0.002f, 0.05f, -lat);
  // acceleration / desired_gap (idm_policy.py:254-271)
  float ratio = fmaxf(speed, 0.0f) / not_zero(r.target, 0.0f);
  float r2 = ratio * ratio, r4 = r2 * r2, r8 = r4 * r4;
  float acc = 1.0f - r8 * r2;
  if (front_obj >= 0) {
    float hx = S.ux[base + s], hy = S.uy[base + s];
    float fsp = S.spd[base + front_obj];
    float dvx = speed * hx - fsp * S.ux[base + front_obj], dvy = speed * hy - fsp * S.uy[base + front_obj];
    float dv = dvx * hx + dvy * hy;
    float d_star = 10.0f + speed * 1.5f + speed * dv / (2.0f * 2.2360679774997896f);
    float sd = d_star / not_zero(front_dist, 1e-2f);
    acc -= sd * sd;
  }
  out_steer = steering;
  out_acc = acc;
  PHASE_MARK(12);  // idm: PID + IDM law
}

// kinematic bicycle (component/highway_vehicle/kinematics.py:134-156) driven by the reference's action -> force mapping
// (base_vehicle.py:343-376); see DESIGN.md §3 for the substitution of Bullet's raycast vehicle.
DEV void dynamics(const PgdDev& d, const pgd_spawn& p, Veh& r, bool reverse) {
  float dt = d.cfg.dt;
  float force = 0.0f, brake = 0.0f;
  if (r.thr >= 0.0f) {
    brake = 2.0f;
    force = (fabsf(r.v) * 3.6f > p.max_speed) ? 0.0f : p.max_engine_force * r.thr;
  } else if (reverse) {  // enable_reverse: engine force backwards, no brake (base_vehicle.py:370-373)
    force = p.max_engine_force * r.thr;
  } else {
    brake = fabsf(r.thr) * p.max_brake_force;
  }
  float delta = -clipf(r.steer, -1.0f, 1.0f) * p.max_steer;
  // beta = atan(t), t = tan(delta)/2  ->  cos(beta) = 1/sqrt(1+t^2), sin(beta) = t/sqrt(1+t^2)
  float t = 0.5f * tanf(delta);
  float cb = 1.0f / sqrtf(1.0f + t * t), sb = t * cb;
  // unit vector of the motion direction th + beta, advanced by exact small-angle rotations instead of sincos per sub-step
  float cd = r.hx * cb - r.hy * sb, sd = r.hy * cb + r.hx * sb;
  float inv_half_base = 2.0f / p.wheelbase;
  float dv_brake = fminf(4.0f * brake / p.mass, p.friction * 9.81f * dt);
  float dv_engine = 4.0f * force / p.mass * dt;
  for (int k = 0; k < d.cfg.decision_repeat; ++k) {
    r.x += r.v * cd * dt;
    r.y += r.v * sd * dt;
    float dth = r.v * sb * inv_half_base * dt;  // |dth| < 0.25 rad at 80 km/h and full lock
    r.th += dth;
    float q = dth * dth;
    float sn = dth * (1.0f + q * (-1.0f / 6.0f + q * (1.0f / 120.0f + q * (-1.0f / 5040.0f))));
    float cs = 1.0f + q * (-0.5f + q * (1.0f / 24.0f + q * (-1.0f / 720.0f + q * (1.0f / 40320.0f))));
    float ncd = cd * cs - sd * sn;
    sd = sd * cs + cd * sn;
    cd = ncd;
    if (force != 0.0f) r.v += dv_engine;
    else r.v = r.v >= 0.0f ? fmaxf(0.0f, r.v - dv_brake) : fminf(0.0f, r.v + dv_brake);
    if (!reverse) r.v = fmaxf(r.v, 0.0f);
  }
  // heading unit vector = motion direction rotated back by beta, renormalised
  float hx = cd * cb + sd * sb, hy = sd * cb - cd * sb;
  float inv = 1.0f / sqrtf(hx * hx + hy * hy);
  r.hx = hx * inv;
  r.hy = hy * inv;
}

DEV void reset_vehicle(const pgd_spawn& p, Veh& r, int spawn_index, bool is_agent) {  // base_vehicle.py:292-339
  memset(&r, 0, sizeof(Veh));
  // agents have no PID state: under PGD_MA_TOLLGATE the fields carry in_toll_time = 0 and entry / exit / last block = none
  // (marl_tollgate.py:36-60,76-96); harmless otherwise
  if (is_agent) { r.php = (float)p.aux; r.phi = -1.0f; r.plp = -1.0f; r.pli = -1.0f; }  // php: parking destination / toll time
  r.spawn = spawn_index;
  r.rlane = is_agent ? 0 : -1;  // agents: episode length; traffic: IDMPolicy.routing_target_lane = None
  r.hx = 1.0f;
  if (p.lane < 0) { r.status = ST_EMPTY; return; }
  r.status = p.group == -1 ? ST_ACTIVE : ST_PENDING;  // PGD_GROUP_NEVER (-2): in the world, never driven
  r.x = p.x; r.y = p.y; r.th = p.heading;
  r.lastx = p.x; r.lasty = p.y;
  sincosf(p.heading, &r.lasthy, &r.lasthx);
  r.hx = r.lasthx; r.hy = r.lasthy;
  r.target = 30.0f;
  r.lane = p.lane;
  r.ck0 = 0;
  r.ck1 = p.n_ckpt > 2 ? 1 : 0;
  r.timer = p.timer0;
}

// reward / done: envs/pgdrive_env.py:162-258, base_vehicle.py:738-745
DEV float reward_done(const PgdDev& d, const MapView& mv, const pgd_spawn& sp, const Veh& r, const RouteCtx& ctx,
                      unsigned& flags_out, bool& done_out) {
  const pgd_config& g = d.cfg;
  unsigned vf = (unsigned)r.vflags;
  const pgd_lane& VL = mv.lanes[r.lane];
  bool in_ref = VL.road == ctx.road_cur;
  const pgd_lane& cl = in_ref ? VL : mv.lanes[ctx.cur_first];
  float positive = (in_ref || (g.marl_flags & PGD_MA_PLAIN_REWARD)) ? 1.0f : (mv.roads[VL.road].negative ? -1.0f : 1.0f);
  float l0, t0, l1, t1;
  lane_local(cl, r.lastx, r.lasty, l0, t0);
  lane_local(cl, r.x, r.y, l1, t1);
  float w = mv.m->lane_width;
  float lateral_factor = g.use_lateral ? clipf(1.0f - 2.0f * fabsf(t1) / w, 0.0f, 1.0f) : 1.0f;
  float reward = g.driving_reward * (l1 - l0) * lateral_factor * positive;
  if (g.marl_flags & PGD_MA_TOLLGATE) {  // MultiAgentTollgateEnv.reward_function (marl_tollgate.py:195-232)
    if (ctx.blk == '$') {
      // BaseVehicle.overspeed (base_vehicle.py:759-761): lane.speed_limit (3 on toll lanes, 1000 elsewhere) < speed [km/h]
      const bool lane_toll = mv.roads[VL.road].block_id == '$';
      if (lane_toll && 3.0f < speed_kmh(r.v)) reward = -g.overspeed_penalty * speed_kmh(r.v) / sp.max_speed;
    } else reward += g.speed_reward * (speed_kmh(r.v) / sp.max_speed);
  } else
  reward += g.speed_reward * (speed_kmh(r.v) / sp.max_speed) * positive;
  unsigned out = vf & (PGD_F_ON_YELLOW | PGD_F_ON_WHITE | PGD_F_ON_BROKEN | PGD_F_CRASH_SIDEWALK | PGD_F_OFF_LANE |
                       PGD_F_OUT_OF_ROUTE | PGD_F_CRASH_VEHICLE | PGD_F_CRASH_OBJECT | PGD_F_CRASH_BUILDING);
  const pgd_lane& fl = mv.lanes[sp.dest_lane];
  float lon, lat;
  lane_local(fl, r.x, r.y, lon, lat);
  bool arrive = (fl.length - 5.0f < lon && lon < fl.length + 5.0f) && (w * 0.5f >= lat && lat >= (0.5f - ctx.cur_n) * w);
  unsigned oor_bits = (g.marl_flags & PGD_MA_TOLLGATE) ? PGD_F_CRASH_SIDEWALK  // marl_tollgate.py:234-240
                      : (g.marl_flags & PGD_MA_PARKING) ? (PGD_F_OFF_LANE | PGD_F_CRASH_SIDEWALK)  // marl_parking_lot.py:213-217
                                                        : (PGD_F_ON_WHITE | PGD_F_OFF_LANE | PGD_F_CRASH_SIDEWALK);
  if (!(g.marl_flags & PGD_MA_YELLOW_OK)) oor_bits |= PGD_F_ON_YELLOW;
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

// ---------------------------------------------------------------------------------------------------------------------
// observation: LidarStateObservation.observe (obs/state_obs.py:132-170) for one (env, agent)
// ---------------------------------------------------------------------------------------------------------------------
DEV void navi_info_for(const pgd_lane& ref, float w, int n_cur, float px, float py, float hx, float hy, float* out) {
  // Navigation._get_info_for_checkpoint (navigation.py:213-260); ref = ref_lanes[0] of the checkpoint's road
  float later_middle = ((float)n_cur * 0.5f - 0.5f) * w;
  float cx, cy;
  lane_position(ref, ref.length, later_middle, cx, cy);
  float dx = cx - px, dy = cy - py;
  float dn = norm2(dx, dy);
  if (dn > 50.0f) { dx = dx / dn * 50.0f; dy = dy / dn * 50.0f; }
  float ph, ps;
  projection(hx, hy, dx, dy, ph, ps);
  float bend = 0.0f, dir = 0.0f, angle = 0.0f;
  if (ref.dir != 0.0f) {
    bend = ref.bx / (60.0f + n_cur * w);
    dir = ref.dir;
    angle = dir == 1.0f ? ref.c - ref.by : ref.by - ref.c;
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

// nearest hit fraction of the segment p + t d (t in [0,1]) against the map's boxes whose kind is in `kinds`: Amanatides-Woo
// walk over the uniform grid; a cell is skipped once its entry parameter is beyond the best hit so far
DEV float ray_grid(const MapView& mv, float px, float py, float dx, float dy, unsigned kinds) {
  const pgd_map& m = *mv.m;
  const float inv = 1.0f / m.cell;
  int ix = (int)floorf((px - m.ox) * inv), iy = (int)floorf((py - m.oy) * inv);
  const int sx = dx > 0.0f ? 1 : -1, sy = dy > 0.0f ? 1 : -1;
  const float big = 3.0e38f;
  const float tdx = dx != 0.0f ? fabsf(m.cell / dx) : big, tdy = dy != 0.0f ? fabsf(m.cell / dy) : big;
  float tmx = dx != 0.0f ? ((m.ox + (ix + (dx > 0.0f ? 1 : 0)) * m.cell) - px) / dx : big;
  float tmy = dy != 0.0f ? ((m.oy + (iy + (dy > 0.0f ? 1 : 0)) * m.cell) - py) / dy : big;
  float best = 1.0f, t_enter = 0.0f;
  for (int it = 0; it < 64; ++it) {
    if (t_enter > best + 0.02f) break;  // boxes are registered with a 5 cm margin: keep a little slack
    if (ix >= 0 && iy >= 0 && ix < m.gx && iy < m.gy) {
      const int cell = iy * m.gx + ix;
      const int k1 = cell_first(mv.cstart[cell + 1]);
      for (int k = cell_mid(mv.cstart[cell]); k < k1; ++k) {
        const pgd_box b = mv.cbox[k];
        if (!((1u << b.kind) & kinds)) continue;
        best = fminf(best, ray_obb(obb_of(b), px, py, dx, dy));
      }
    } else if ((sx > 0 ? ix >= m.gx : ix < 0) || (sy > 0 ? iy >= m.gy : iy < 0)) {
      break;  // left the grid for good
    }
    if (tmx < tmy) { t_enter = tmx; tmx += tdx; ix += sx; }
    else { t_enter = tmy; tmy += tdy; iy += sy; }
    if (t_enter > 1.0f) break;
  }
  return best;
}

struct ObsLds {  // bodies inside the lidar broad phase of the observing agent, compacted
  float bx[MAXV], by[MAXV], bux[MAXV], buy[MAXV], bhl[MAXV], bhw[MAXV], bspd[MAXV];
  float bdist[MAXV];  // centre distance; +inf for traffic objects, which are never ranked as neighbour vehicles
  int n, nveh;
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
template <bool OBJ>
DEV void obs_compact(ObsLds& L, int o, int a, bool present, bool is_vehicle, float x, float y, float ux, float uy, float hl,
                     float hw, float spd, float px, float py, float R) {
  if (!OBJ) is_vehicle = true;
  bool in = present && o != a && shape_point_dist<OBJ>(Obb{x, y, ux, uy, hl, hw}, px, py) <= R;
  unsigned long long m = __ballot(in), mv_ = OBJ ? __ballot(in && is_vehicle) : m;
  if (in) {
    int k = __popcll(m & ((1ull << o) - 1ull));
    L.bx[k] = x; L.by[k] = y; L.bux[k] = ux; L.buy[k] = uy; L.bhl[k] = hl; L.bhw[k] = hw; L.bspd[k] = spd;
    L.bdist[k] = is_vehicle ? norm2(px - x, py - y) : __builtin_inff();
  }
  if (o == 0) { L.n = __popcll(m); L.nveh = __popcll(mv_); }
}

// writes the D floats of one agent's row with `nt` cooperating threads (tid in [0, nt))
template <bool OBJ>
DEV void observe_agent(const PgdDev& d, const MapView& mv, const pgd_spawn& sp, const AgentView& ag, const ObsLds& L,
                       float* __restrict__ row, int tid, int nt) {
  const float px = ag.x, py = ag.y, hx = ag.hx, hy = ag.hy;
  const float R = d.cfg.lidar_dist;
  const int NL = d.cfg.num_lasers;
  // StateObservation.vehicle_state (state_obs.py:58-106) + navi info (navigation.py:185-197): one lane per float.
  // Row layout: [side fan k | 2 lateral distances][6 ego floats][lane-line fan m][10 navi][4*NO neighbours][NL beams]
  const int KS = d.cfg.side_lasers, KM = d.cfg.lane_line_lasers;
  const bool toll = (d.cfg.marl_flags & PGD_MA_TOLLGATE) != 0;  // no navigation block, 2 toll floats after the lidar
  const int RAM = d.cfg.random_agent_model ? 2 : 0;  // LENGTH / 10, WIDTH / 2.5 after the lane-line fan (state_obs.py:102-105)
  const int o_ego = KS > 0 ? KS : 2, o_navi = o_ego + 6 + KM + RAM, o_oth = o_navi + (toll ? 0 : 10);
  if (RAM && tid == nt - 1) {
    row[o_ego + 6 + KM] = clipf(sp.length / 10.0f, 0.0f, 1.0f);
    row[o_ego + 6 + KM + 1] = clipf(sp.width / 2.5f, 0.0f, 1.0f);
  }
  if (toll && tid == 0) {  // TollGateObservation.observe (marl_tollgate.py:84-96)
    const bool in_toll = ag.blk == '$';
    float* t2 = row + o_oth + 4 * d.cfg.num_others + NL;
    t2[0] = in_toll ? 1.0f : 0.0f;
    t2[1] = (in_toll && ag.toll_time > (float)d.cfg.min_pass_steps) ? 1.0f : 0.0f;
  }
  if (tid < 18) {
    // every lane fetches the one lane record its float needs BEFORE the branch ladder, so the reads overlap instead of
    // queueing behind each other branch by branch: heading_diff -> last lane of the current road; navi -> first lanes
    const int lid = tid < 8 ? ag.cur_first + ag.cur_n - 1 : (tid < 13 ? ag.cur_first : ag.next_first);
    const pgd_lane ml = mv.lanes[lid];
    const float max_speed = sp.max_speed;
    float v = 0.0f;
    int col = -1;
    if (tid == 0) { v = clipf(ag.dl / 18.0f, 0.0f, 1.0f); col = KS > 0 ? -1 : 0; }  // (MAX_LANE_NUM+1)*MAX_LANE_WIDTH
    else if (tid == 1) { v = clipf(ag.dr / 18.0f, 0.0f, 1.0f); col = KS > 0 ? -1 : 1; }
    else if (tid == 2) { v = heading_diff(ml, px, py, hx, hy); col = o_ego; }
    else if (tid == 3) { v = clipf((speed_kmh(ag.v) + 1.0f) / (max_speed + 1.0f), 0.0f, 1.0f); col = o_ego + 1; }
    else if (tid == 4) { v = clipf((ag.steer / 60.0f + 1.0f) * 0.5f, 0.0f, 1.0f); col = o_ego + 2; }
    else if (tid == 5) { v = clipf((ag.a0s + 1.0f) * 0.5f, 0.0f, 1.0f); col = o_ego + 3; }
    else if (tid == 6) { v = clipf((ag.a0t + 1.0f) * 0.5f, 0.0f, 1.0f); col = o_ego + 4; }
    else if (tid == 7) {
      // acos(clip(cos_beta, 0, 1)) (state_obs.py:87-92) evaluated as atan2(|cross|, dot): identical for unit vectors,
      // but well-conditioned in fp32 near beta = 0 where 1 - cos(beta) underflows the mantissa
      float dot = hx * ag.lhx + hy * ag.lhy, cross = hx * ag.lhy - hy * ag.lhx;
      float beta = dot <= 0.0f ? 0.5f * PGD_PI : atan2f(fabsf(cross), dot);
      v = clipf(beta / 0.1f, 0.0f, 1.0f);
      col = o_ego + 5;
    } else {  // lanes 8..12 -> checkpoint 1, 13..17 -> checkpoint 2
      int which = (tid - 8) / 5, comp = (tid - 8) - which * 5;
      float out[5];
      navi_info_for(ml, mv.m->lane_width, ag.cur_n, px, py, hx, hy, out);
      v = comp == 0 ? out[0] : comp == 1 ? out[1] : comp == 2 ? out[2] : comp == 3 ? out[3] : out[4];
      col = toll ? -1 : o_navi + (tid - 8);
    }
    if (col >= 0) row[col] = v;
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
    row[side ? i : o_ego + 6 + i] = ray_grid(mv, px, py, dist * cs, dist * sn, kinds);
  }
  PHASE_MARK(22);  // obs: state + navi block
  if (NL <= 0) return;
  // get_surrounding_vehicles_info (lidar.py:55-77): rank by centre distance (stable), 4 floats per neighbour; the last
  // threads take this part so that it overlaps the state block of the first ones
  const int NO = d.cfg.num_others;
  const int n = L.n, nveh = OBJ ? L.nveh : n;
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
  // lidar (distance_detector.py:65-94, cutils.pyx:60-142): beam i at theta + i*2pi/N, nearest hit fraction
  const float unit = 2.0f * PGD_PI / (float)NL;
  for (int i = tid; i < NL; i += nt) {
    float ang = (float)i * unit + ag.th;
    float sn, cs;
    sincosf(ang, &sn, &cs);
    float dx = R * cs, dy = R * sn;
    float best = 1.0f;
    for (int k = 0; k < n; ++k)
      best = fminf(best, shape_ray<OBJ>(Obb{L.bx[k], L.by[k], L.bux[k], L.buy[k], L.bhl[k], L.bhw[k]}, px, py, dx, dy));
    if (d.cfg.lidar_gaussian_noise > 0.0f || d.cfg.lidar_dropout_prob > 0.0f) {  // state_obs.py:172-182
      const uint32_t key = 0x51d0a000u + (uint32_t)ag.slot * 1024u + (uint32_t)i;
      if (d.cfg.lidar_gaussian_noise > 0.0f) {
        const float u1 = ((float)(pgd_rng(d.cfg.seed, (uint32_t)ag.env, key, ag.tick) >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(pgd_rng(d.cfg.seed ^ 0x9e3779b9u, (uint32_t)ag.env, key, ag.tick) >> 8) + 0.5f) * (1.0f / 16777216.0f);
        best = clipf(best + d.cfg.lidar_gaussian_noise * sqrtf(-2.0f * logf(u1)) * cosf(2.0f * PGD_PI * u2), 0.0f, 1.0f);
      }
      if (d.cfg.lidar_dropout_prob > 0.0f) {
        const float u3 = ((float)(pgd_rng(d.cfg.seed ^ 0x7f4a7c15u, (uint32_t)ag.env, key, ag.tick) >> 8) + 0.5f) * (1.0f / 16777216.0f);
        if (u3 < d.cfg.lidar_dropout_prob) best = 0.0f;
      }
    }
    row[o_oth + 4 * NO + i] = best;
  }
  PHASE_MARK(24);  // obs: lidar
}

// ---------------------------------------------------------------------------------------------------------------------
// k_step: one env.step() for every environment (base_env.py:184-224)
// lane -> (group g = lane / SUB, sub-lane); group g -> (env-local el = g / V, slot s = g % V)
// ---------------------------------------------------------------------------------------------------------------------
// cache warm-up load: one dword of the line at `p` goes straight to LDS (no VGPR destination, nothing waits for it)
DEV void touch_line(const void* p, int* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                   (__attribute__((address_space(3))) void*)lds, 4, 0, 0);
}

struct LaneMap {
  int sub, lead, el, s, e, idx, base;
  bool valid;
};
DEV LaneMap lane_map(const PgdDev& d, int unit, int n_units) {
  LaneMap m;
  int lane = threadIdx.x;
  int g = lane / d.sub;
  m.sub = lane - g * d.sub;
  m.lead = g * d.sub;
  m.el = g / d.V;
  m.s = g - m.el * d.V;
  m.e = unit * d.epw + m.el;
  m.valid = (m.el < d.epw) && (m.e < n_units);
  m.base = m.el * d.V;
  m.idx = m.e * d.V + m.s;
  return m;
}

extern __shared__ __align__(16) unsigned char s_dyn[];  // [lanes | roads] of the block's map when epw == 1

#define FUSE_MAX_AGENTS 8
#ifndef PGD_WAVES_PER_SIMD
#define PGD_WAVES_PER_SIMD 4  // <=128 VGPRs: 4096 envs = 4096 waves are then all resident at once (16 per CU)
#endif
// ONE_ENV (epw == 1: every lane of the wave works on env blockIdx.x) is a compile-time switch: the env index, its
// scenario, the map view (7 table pointers) and the env counters are then wave-uniform and live in SGPRs instead of
// occupying ~20 VGPRs per lane for the whole kernel.
// MARL (multi-agent tail: delay-done, respawn, __all__) is compiled in only for the multi-agent engine.
// OBJ: traffic objects present in some scenario (circle shapes, crash_object bookkeeping).
template <bool ONE_ENV, bool MARL, bool OBJ>
__global__ __launch_bounds__(WAVE, PGD_WAVES_PER_SIMD) void k_step(PgdDev d, const float* __restrict__ act, float* __restrict__ reward,
                                                uint8_t* __restrict__ done, uint32_t* __restrict__ flags,
                                                float* __restrict__ obs) {
  __shared__ Snap S;
  __shared__ ObsLds OL;
  __shared__ AgentView s_ag[FUSE_MAX_AGENTS];
  __shared__ int s_flag[WAVE];
  __shared__ int s_hit[WAVE];  // per snapshot slot: an agent's chassis overlaps another vehicle
  __shared__ int s_pf[WAVE];   // landing zone of the warm-up loads
  __shared__ int s_aux;        // multi-agent parking lot: pool of free parking spaces (bit mask)
  __shared__ int s_kind[WAVE]; // PGD_OBJ_* of every slot (fused observation: objects are lidar targets, not neighbours)
  const int V = d.V, A = d.A, N = d.N;
  const int lane = threadIdx.x;
  const LaneMap lm = lane_map(d, blockIdx.x, N);
  const int el = ONE_ENV ? 0 : lm.el, s = lm.s, e = ONE_ENV ? (int)blockIdx.x : lm.e, base = ONE_ENV ? 0 : lm.base;
  const bool valid = lm.valid, leader = lm.sub == 0;
  const Grp g{lm.sub, d.sub, lm.lead};
  const int slot = base + s;  // my entry of the LDS snapshot

  PHASE_INIT();
  Veh r;
  RouteCtx ctx{0, 0, 0, 1, 0};  // of this lane's vehicle if it is an agent: refreshed by every after_step_vehicle
  MapView mv;
  const pgd_spawn* sp = nullptr;
  const pgd_scenario* sc = nullptr;
  int ng = 0, ep_steps = 0;
  uint32_t steps_total = 0;
  S.present[lane] = 0;
  s_flag[lane] = 0;
  s_hit[lane] = 0;
  constexpr bool one_env = ONE_ENV;
  constexpr bool marl = MARL;
  int scen = 0;
  // the vehicle record does not depend on the scenario: its 8 x 16 B reads go out first and overlap the scalar chain
  // env counters -> scenario -> map header -> table pointers below
  if (valid) load_veh(d, e, s, r);
  if (one_env || valid) {
    scen = d.ei[(size_t)e * PGD_NEI + EI_SCEN];
    sc = d.scen + scen;
    mv = map_view_of(d, d.scen_map + scen);
    ng = d.ei[(size_t)(e) * PGD_NEI + EI_NEXT_GROUP];
    ep_steps = d.ei[(size_t)(e) * PGD_NEI + EI_EP_STEPS];
    steps_total = (uint32_t)d.ei[(size_t)(e) * PGD_NEI + EI_STEPS_TOTAL];
  }
  if (one_env && d.lds_bytes > 0) {
    // stage the env's lane + road tables in LDS (coalesced 16 B loads by all 64 lanes)
    const pgd_map* bm = mv.m;
    const int nl16 = bm->n_lanes * 4, nr16 = bm->n_roads;  // 16-byte units
    if ((nl16 + nr16) * 16 <= d.lds_bytes) {
      const uint4* gl = reinterpret_cast<const uint4*>(mv.lanes);
      const uint4* gr = reinterpret_cast<const uint4*>(mv.roads);
      uint4* sl = reinterpret_cast<uint4*>(s_dyn);
      const int n16 = nl16 + nr16;  // roads follow the lanes in LDS; both source ranges are 16 B aligned
      for (int k = lane; k < n16; k += 4 * WAVE) {
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int kk = k + j * WAVE;
          if (kk < n16) v[j] = kk < nl16 ? gl[kk] : gr[kk - nl16];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int kk = k + j * WAVE;
          if (kk < n16) sl[kk] = v[j];
        }
      }
      mv.lanes = reinterpret_cast<const pgd_lane*>(s_dyn);
      mv.roads = reinterpret_cast<const pgd_road*>(s_dyn + (size_t)bm->n_lanes * sizeof(pgd_lane));
    }
  }
#ifdef PGD_WARM
  if (ONE_ENV) {
    // L2 warm-up: the L2 starts cold at every launch; touch the env's tables now (global -> LDS loads without a VGPR
    // destination) so that the dependent lookups of the later phases hit in L2
    const pgd_map* bm = mv.m;
    const char* lp = reinterpret_cast<const char*>(mv.lanes);
    const int nl = (bm->n_lanes * 64 + 127) / 128;
    for (int k = lane; k < nl; k += WAVE) touch_line(lp + (size_t)k * 128, s_pf);
    const char* rp = reinterpret_cast<const char*>(mv.roads);
    const int nr = (bm->n_roads * 16 + 127) / 128;
    if (lane < nr) touch_line(rp + (size_t)lane * 128, s_pf);
    const char* spp = reinterpret_cast<const char*>(d.spawns + (size_t)scen * d.sstride);
    const int ns = (V * (int)sizeof(pgd_spawn) + 127) / 128;
    if (lane < ns) touch_line(spp + (size_t)lane * 128, s_pf);
  }
#endif
  PHASE_MARK(13);  // load: scenario + table staging
  if (valid) {
    sp = d.spawns + (size_t)scen * d.sstride + r.spawn;
    // (0) AgentManager.before_step (agent_manager.py:191-199): finished agents count down, then leave the world
    if (marl && r.status == ST_DYING && --r.timer == 0) r.status = ST_EMPTY;
  }
  __syncthreads();
  if (valid) {
    // (1) TrafficManager.before_step trigger (traffic_manager.py:76-85)
    if (s < A && r.status == ST_ACTIVE && ng < sc->n_groups && mv.lanes[r.lane].road == sc->trigger_road[ng]) s_flag[el] = 1;
  }
  __syncthreads();
  PHASE_MARK(0);  // load
  const bool trig = ONE_ENV ? (__ballot(s_flag[0] != 0) != 0ull) : (valid && s_flag[el] != 0);
  if (valid && trig && r.status == ST_PENDING && sp->group == ng) r.status = ST_ACTIVE;
  if (trig) ng += 1;  // every lane of the env keeps the same copy
  // snapshot of the world before physics
  if (valid) {
    if (leader) {
      S.x[slot] = r.x; S.y[slot] = r.y; S.ux[slot] = r.hx; S.uy[slot] = r.hy;
      S.spd[slot] = speed_kmh(r.v);
      const int kind = OBJ ? (int)sp->kind : PGD_OBJ_VEHICLE;
      if (OBJ) s_kind[slot] = kind;
      S.hl[slot] = 0.5f * sp->length; S.hw[slot] = kind == PGD_OBJ_CYLINDER ? -1.0f : 0.5f * sp->width;
      S.lane[slot] = r.lane;
      const bool present = r.status == ST_PENDING || r.status == ST_ACTIVE || r.status == ST_DYING;
      S.present[slot] = present ? 1 : 0;
      if (present && V > A) {
        const pgd_lane& ml = mv.lanes[r.lane];
        float lo, la;
        lane_local(ml, r.x, r.y, lo, la);
        S.lon[slot] = lo;
        S.llen[slot] = ml.length;
        S.succ[slot] = *reinterpret_cast<const int4*>(ml.succ);
      }
    }
  }
  __syncthreads();
  PHASE_MARK(1);  // trigger + snapshot
  // slots present in the world: one ballot when the wave carries one env (lane o reads slot o), else "all, test later"
  const unsigned long long pmask = ONE_ENV ? __ballot(lane < V && S.present[lane] != 0) : ((V >= 64 ? 0ull : (1ull << V)) - 1ull);
  const bool acting = valid && r.status == ST_ACTIVE;
  // (2) policies
  if (acting) {
    float st, tb;
    if (s < A) {  // EnvInputPolicy.act (env_input_policy.py:17-26); NaN made harmless (test_ego_vehicle.py:78-84)
      float a0 = act[((size_t)e * A + s) * 2 + 0], a1 = act[((size_t)e * A + s) * 2 + 1];
      if (a0 != a0) a0 = 0.0f;
      if (a1 != a1) a1 = 0.0f;
      st = clipf(a0, -1.0f, 1.0f);
      tb = clipf(a1, -1.0f, 1.0f);
      if (d.cfg.discrete_action) {  // convert_to_continuous_action on the CLIPPED action (env_input_policy.py:17-31)
        st = st * (2.0f / (float)(d.cfg.discrete_steering_dim - 1)) - 1.0f;
        tb = tb * (2.0f / (float)(d.cfg.discrete_throttle_dim - 1)) - 1.0f;
      }
    } else {
      idm_act<OBJ>(d, mv, g, *sp, S, base, V, s, e, steps_total, pmask, r, st, tb);
    }
    PHASE_MARK(2);  // policy (IDM)
    // (3) BaseVehicle.before_step (base_vehicle.py:238-253)
    r.vflags &= ~(PGD_F_CRASH_VEHICLE | PGD_F_CRASH_OBJECT | PGD_F_CRASH_BUILDING);
    r.lastx = r.x; r.lasty = r.y;
    r.lasthx = r.hx; r.lasthy = r.hy;
    r.a0s = r.a1s; r.a0t = r.a1t;
    r.a1s = st; r.a1t = tb;
    // _set_action / _set_incremental_action (base_vehicle.py:343-358)
    r.steer = (s < A && d.cfg.increment_steering) ? clipf(r.steer + st * 0.05f, -1.0f, 1.0f) : st;
    r.thr = tb;
    // (4) physics
    dynamics(d, *sp, r, s < A && d.cfg.enable_reverse != 0);
    PHASE_MARK(3);  // dynamics
  }
  __syncthreads();
  if (acting && leader) { S.x[slot] = r.x; S.y[slot] = r.y; S.ux[slot] = r.hx; S.uy[slot] = r.hy; }
  __syncthreads();
  // (5) vehicle-vehicle contacts on the post-physics poses (collision_callback.py:7-36): every body in the world tests
  // itself against each agent of its env, so the A x V pair tests run in parallel lanes
  if (!OBJ) {
    if (valid && leader && S.present[slot]) {
      Obb me = snap_obb(S, slot);
      for (int a = 0; a < A; ++a)
        if (a != s && obb_overlap(snap_obb(S, base + a), me)) s_hit[base + a] = 1;
    }
  } else {
    const int my_kind = valid ? s_kind[slot] : PGD_OBJ_VEHICLE;
    if (valid && S.present[slot] && (leader || my_kind != PGD_OBJ_VEHICLE)) {  // object sub-lanes all keep their copy of the bit
      const Obb me = snap_obb(S, slot);
      const int kind = my_kind;
      // a traffic object reports only its first contact (TrafficObject.crashed / COST_ONCE, collision_callback.py:27-32)
      const bool live = kind == PGD_OBJ_VEHICLE || !(r.vflags & (int)PGD_F_OBJECT_HIT);
      bool touched = false;
      for (int a = 0; a < A; ++a)
        if (a != s && S.present[base + a] && shape_overlap<true>(snap_obb(S, base + a), me)) {
          touched = true;
          if (leader && live) atomicOr(&s_hit[base + a], kind == PGD_OBJ_VEHICLE ? 1 : (kind == PGD_OBJ_BUILDING ? 4 : 2));
        }
      if (touched && kind != PGD_OBJ_VEHICLE && kind != PGD_OBJ_BUILDING) r.vflags |= (int)PGD_F_OBJECT_HIT;  // all sub-lanes
    }
  }
  __syncthreads();
  if (acting && s < A) {
    if (s_hit[slot] & 1) r.vflags |= PGD_F_CRASH_VEHICLE;
    if (OBJ && (s_hit[slot] & 2)) r.vflags |= PGD_F_CRASH_OBJECT;
    if (OBJ && (s_hit[slot] & 4)) r.vflags |= PGD_F_CRASH_BUILDING;
  }
  PHASE_MARK(4);  // crash
  // (6) after_step; traffic off the lanes is removed (traffic_manager.py:91-109)
  if (acting) {
    after_step_vehicle(mv, g, *sp, r, s < A, !one_env, ctx);
    if (s >= A && (r.vflags & PGD_F_OFF_LANE)) r.status = ST_REMOVED;
  }
  if (one_env) {  // line / sidewalk test of each agent by the whole wave (base_vehicle.py:615-644)
    if (leader && valid && s < A) s_flag[A + s] = acting ? 1 : 0;
    __syncthreads();
    for (int a = 0; a < A; ++a) {
      if (!s_flag[A + a]) continue;
      unsigned fl = state_check_wave(mv, snap_obb(S, a));
      if (valid && s == a) r.vflags |= (int)fl;
    }
  }
  PHASE_MARK(5);  // after_step
  ep_steps += 1;
  steps_total += 1;
  // (7) reward / done (base_env.py:303-344)
  s_flag[lane] = 0;
  __syncthreads();
  unsigned my_fl = 0;
  bool my_dn = false;
  float my_rew = 0.0f;
  const bool was_active = acting;  // status at the start of the step (after the delay-done countdown)
  if (valid && s < A && !marl) {
    if (r.status == ST_ACTIVE) my_rew = reward_done(d, mv, *sp, r, ctx, my_fl, my_dn);
    if (d.cfg.horizon > 0 && ep_steps >= d.cfg.horizon) { my_dn = true; my_fl |= PGD_F_MAX_STEP; }
    if (sc->max_steps > 0 && ep_steps >= sc->max_steps) { my_dn = true; my_fl |= PGD_F_MAX_STEP; }  // auto_termination
    r.eprew += my_rew;
    bool will_reset = my_dn && d.cfg.auto_reset && A == 1;
    if (will_reset) { my_fl |= PGD_F_RESET; s_flag[el] = 1; }
  }
  if (marl && one_env) {
    // ---- multi-agent tail: multi_agent_pgdrive.py:109-213, agent_manager.py:134-175, spawn_manager.py:160-215 ----
    const pgd_config& gcf = d.cfg;
    const bool toll = (gcf.marl_flags & PGD_MA_TOLLGATE) != 0;
    const bool parking = (gcf.marl_flags & PGD_MA_PARKING) != 0;
    if (parking && lane == 0) s_aux = d.ei[(size_t)blockIdx.x * PGD_NEI + EI_AUX];  // parking_space_available
    if (parking) __syncthreads();
    if (valid && s < A && was_active) {
      if (toll && ctx.blk == '$') r.php += 1.0f;  // TollGateObservation.observe counts its calls inside the toll block
      my_rew = reward_done(d, mv, *sp, r, ctx, my_fl, my_dn);
      const bool arrive = my_fl & PGD_F_ARRIVE, oor = my_fl & PGD_F_OUT_OF_ROAD, crash = my_fl & PGD_F_CRASH_VEHICLE;
      if (crash && !(gcf.marl_flags & PGD_MA_CRASH_DONE) && !(arrive || oor)) my_dn = false;
      if (oor && !(gcf.marl_flags & PGD_MA_OUT_ROAD_DONE) && !arrive) my_dn = false;
      if (toll && r.phi >= 0.0f && r.plp >= 0.0f && r.plp - r.phi < (float)gcf.min_pass_steps) {  // marl_tollgate.py:262-268
        my_dn = true;
        my_fl |= PGD_F_OUT_OF_ROAD;
      }
      r.rlane += 1;  // episode_length
      if (gcf.horizon > 0 && r.rlane >= gcf.horizon) { my_dn = true; my_fl |= PGD_F_MAX_STEP; }
      r.eprew += my_rew;
      my_fl |= PGD_F_REPORT;
      if (my_dn && parking && r.php > 0.0f) {  // ParkingLotSpawnManager.after_vehicle_done: its space is free again
        if (leader) atomicOr(&s_aux, 1 << ((int)r.php - 1));
        r.php = 0.0f;
      }
      if (my_dn) {  // AgentManager.finish
        if (arrive || gcf.delay_done <= 0) r.status = ST_EMPTY;
        else { r.status = ST_DYING; r.timer = gcf.delay_done; }
      }
    }
    // the world after the finishes (leaders publish, everybody reads)
    __syncthreads();
    if (valid && leader) {
      S.x[slot] = r.x; S.y[slot] = r.y; S.ux[slot] = r.hx; S.uy[slot] = r.hy;
      S.hl[slot] = 0.5f * sp->length; S.hw[slot] = 0.5f * sp->width;
      S.present[slot] = (r.status == ST_PENDING || r.status == ST_ACTIVE || r.status == ST_DYING) ? 1 : 0;
    }
    __syncthreads();
    const bool is_lead_agent = valid && leader && s < A;
    int alive = __popcll(__ballot(is_lead_agent && (r.status == ST_ACTIVE || r.status == ST_DYING)));
    int next_agent = d.ei[(size_t)blockIdx.x * PGD_NEI + EI_NEXT_AGENT];
    const bool allow = (gcf.marl_flags & PGD_MA_ALLOW_RESPAWN) && !(gcf.horizon > 0 && ep_steps >= gcf.horizon) &&
                       alive < gcf.agent_limit;
    if (allow) {
      const pgd_spawn* rbase = d.spawns + (size_t)scen * d.sstride + V;
      for (int p = 0; p < gcf.respawn_places; ++p) {
        const pgd_spawn& place = rbase[p * gcf.respawn_dests];
        float ps, pc;
        sincosf(place.heading, &ps, &pc);
        const Obb region{place.x, place.y, pc, ps, 4.0f, 1.5f};  // RESPAWN_REGION 8 m x 3 m (spawn_manager.py:27-28)
        const bool blocks = lane < V && S.present[lane] && obb_overlap(region, snap_obb(S, lane));
        if (__ballot(blocks) != 0ull) continue;
        // lowest empty slot that did not report this step (its terminal row must survive)
        unsigned long long em = __ballot(is_lead_agent && r.status == ST_EMPTY && !(my_fl & PGD_F_REPORT));
        if (em == 0ull) break;
        const int src = __builtin_ffsll((long long)em) - 1;
        const int tslot = (src / d.sub) % V;
        int dest = (int)(pgd_rng(gcf.seed, (uint32_t)blockIdx.x, 0x0a9e47u + (uint32_t)p, (uint32_t)next_agent) %
                         (uint32_t)gcf.respawn_dests);
        if (parking) {  // get_parking_space: a random one of the free spaces; none -> nobody enters from a road
          const unsigned mask = (unsigned)s_aux & ((1u << gcf.respawn_dests) - 1u);
          if (__ballot(mask != 0u) == 0ull) break;
          int pick = (int)(pgd_rng(gcf.seed, (uint32_t)blockIdx.x, 0x0a9e47u + (uint32_t)p, (uint32_t)next_agent) %
                           (uint32_t)__popc(mask));
          dest = 0;
          for (int b = 0; b < 32; ++b)
            if (mask & (1u << b)) { if (pick-- == 0) { dest = b; break; } }
          __syncthreads();  // everybody has read the pool before lane 0 takes the space out
          if (lane == 0) s_aux &= ~(1 << dest);
        }
        if (valid && s == tslot) {
          const int sidx = V + p * gcf.respawn_dests + dest;
          sp = d.spawns + (size_t)scen * d.sstride + sidx;
          reset_vehicle(*sp, r, sidx, true);
          r.agent_id = (float)next_agent;
          after_step_vehicle(mv, g, *sp, r, true, true, ctx);
          my_fl |= PGD_F_NEW;
          if (leader) {
            S.x[slot] = r.x; S.y[slot] = r.y; S.ux[slot] = r.hx; S.uy[slot] = r.hy;
            S.hl[slot] = 0.5f * sp->length; S.hw[slot] = 0.5f * sp->width;
            S.present[slot] = 1;
          }
        }
        next_agent += 1;
        __syncthreads();
      }
    }
    // StayTimeManager.record(active_agents, episode_steps) after the step (marl_tollgate.py:36-60,276-279)
    if (toll && valid && s < A && r.status == ST_ACTIVE) {
      const float cur = (float)ctx.blk, last = r.pli;
      r.pli = cur;
      if (last >= 0.0f && last != cur) {
        if (ctx.blk == '$') r.phi = (float)ep_steps;
        else if ((ctx.blk == 'y' || ctx.blk == 'Y') && last == (float)'$') r.plp = (float)ep_steps;
      }
    }
    // d["__all__"] (multi_agent_pgdrive.py:142-148)
    const int n_active = __popcll(__ballot(is_lead_agent && r.status == ST_ACTIVE));
    const bool all_done = n_active == 0 || (gcf.horizon > 0 && ep_steps >= 5 * gcf.horizon);
    if (all_done) {
      my_fl |= PGD_F_ALL_DONE;
      if (gcf.auto_reset) { my_fl |= PGD_F_RESET; s_flag[el] = 1; }
    }
    if (lane == 0) d.ei[(size_t)blockIdx.x * PGD_NEI + EI_NEXT_AGENT] = next_agent;
    if (parking) {
      __syncthreads();
      if (lane == 0) d.ei[(size_t)blockIdx.x * PGD_NEI + EI_AUX] = s_aux;
    }
  }
  __syncthreads();
  PHASE_MARK(6);  // reward/done
  // (8) auto reset (base_env.py:269-301): the whole env restarts from its (possibly re-drawn) scenario
  int episodes = 0;
  // ONE_ENV: s_flag[0] is the env's reset flag, the same for every lane: a scalar branch keeps scen / mv in SGPRs
  const bool resetting = ONE_ENV ? (__ballot(s_flag[0] != 0) != 0ull) : (valid && s_flag[el]);
  if (resetting) {
    episodes = d.ei[(size_t)(e) * PGD_NEI + EI_EPISODES] + 1;
    if (d.cfg.resample_scenario)
      scen = (int)(pgd_rng(d.cfg.seed, (uint32_t)e, 0x5ce9a210u, (uint32_t)episodes) % (uint32_t)d.n_scen);
    sc = d.scen + scen;
    mv = map_view(d, sc->map);  // global tables: the staged map may not be the new one
    ng = 0;
    ep_steps = 0;
  }
  if (valid && resetting) {
    sp = d.spawns + (size_t)scen * d.sstride + s;
    reset_vehicle(*sp, r, s, s < A);
    if (r.status != ST_EMPTY) after_step_vehicle(mv, g, *sp, r, s < A, true, ctx);
    // agent ids restart at 0: id = number of spawned agent slots below this one (agent_manager.py:91-132)
    const unsigned long long am = __ballot(leader && s < A && r.status == ST_ACTIVE);
    if (s < A && r.status == ST_ACTIVE) {
      r.agent_id = A == 1 ? 0.0f : (float)__popcll(am & ((1ull << lm.lead) - 1ull));  // A > 1 implies one env per wave
      if (marl) my_fl |= PGD_F_NEW;
    }
    if (s == 0 && leader) {
      d.ei[(size_t)(e) * PGD_NEI + EI_SCEN] = scen;
      d.ei[(size_t)(e) * PGD_NEI + EI_EPISODES] = episodes;
      d.ei[(size_t)(e) * PGD_NEI + EI_NEXT_AGENT] = A == 1 ? 1 : __popcll(am);
      d.ei[(size_t)(e) * PGD_NEI + EI_AUX] = sc->aux;  // parking: the pool of the new episode
    }
  }
  if (valid && leader && s < A) {
    size_t k = (size_t)e * A + s;
    reward[k] = my_rew;
    done[k] = my_dn ? 1 : 0;
    flags[k] = my_fl;
  }
  PHASE_MARK(7);  // reset
  if (valid && leader) {
    store_veh(d, e, s, r);
    if (s == 0) {
      d.ei[(size_t)(e) * PGD_NEI + EI_NEXT_GROUP] = ng;
      d.ei[(size_t)(e) * PGD_NEI + EI_EP_STEPS] = ep_steps;
      d.ei[(size_t)(e) * PGD_NEI + EI_STEPS_TOTAL] = (int)steps_total;
    }
  }
  PHASE_MARK(8);  // store
  // (9) observation of the new state, fused: the wave already holds every vehicle of the env (obs/state_obs.py:132-170)
  if (ONE_ENV && obs != nullptr) {  // host passes obs only when one_env && A <= FUSE_MAX_AGENTS
    __syncthreads();
    if (valid && leader) {
      const bool present = r.status == ST_PENDING || r.status == ST_ACTIVE || r.status == ST_DYING;
      S.x[slot] = r.x; S.y[slot] = r.y; S.ux[slot] = r.hx; S.uy[slot] = r.hy;
      S.spd[slot] = r.status == ST_DYING ? 0.0f : speed_kmh(r.v);
      S.hl[slot] = 0.5f * sp->length;  // the scenario may have changed on reset
      S.hw[slot] = (OBJ && sp->kind == PGD_OBJ_CYLINDER) ? -1.0f : 0.5f * sp->width;
      if (OBJ) s_kind[slot] = sp->kind;
      S.present[slot] = present ? 1 : 0;
      if (s < A) {
        AgentView& ag = s_ag[s];
        ag.x = r.x; ag.y = r.y; ag.th = r.th; ag.hx = r.hx; ag.hy = r.hy; ag.dl = r.dl; ag.dr = r.dr; ag.v = r.v;
        ag.steer = r.steer; ag.a0s = r.a0s; ag.a0t = r.a0t; ag.lhx = r.lasthx; ag.lhy = r.lasthy;
        ag.cur_first = ctx.cur_first; ag.cur_n = ctx.cur_n; ag.next_first = ctx.next_first;
        ag.blk = ctx.blk; ag.toll_time = r.php;
        ag.env = e; ag.slot = s; ag.tick = steps_total;
      }
    }
    __syncthreads();
    // only launched with one env per wave: `scen` / `mv` are wave-uniform and already those of the new episode after a reset
    const int scen_now = scen;
    const MapView& mvo = mv;
    PHASE_MARK(20);  // obs: publish
    for (int a = 0; a < A; ++a) {
      const AgentView ag = s_ag[a];
      const bool have = lane < V && d.cfg.num_lasers > 0;
      obs_compact<OBJ>(OL, lane, a, have && S.present[lane], OBJ ? (have && s_kind[lane] == PGD_OBJ_VEHICLE) : true, S.x[lane], S.y[lane],
                  S.ux[lane], S.uy[lane], S.hl[lane], S.hw[lane], S.spd[lane], ag.x, ag.y, d.cfg.lidar_dist);
      __syncthreads();
      PHASE_MARK(21);  // obs: compaction
      observe_agent<OBJ>(d, mvo, d.spawns[(size_t)scen_now * d.sstride + a], ag, OL, obs + ((size_t)blockIdx.x * A + a) * d.D, lane,
                    WAVE);
      __syncthreads();
    }
  }
  PHASE_MARK(14);  // fused observation
  PHASE_END();
}

// reset of selected envs (base_env.py:269-301); same lane mapping as k_step, unit = position in the id list
__global__ __launch_bounds__(WAVE) void k_reset(PgdDev d, const int32_t* __restrict__ env_ids,
                                                 const int32_t* __restrict__ scen_ids, int n) {
  const int V = d.V, A = d.A, N = d.N;
  const LaneMap lm = lane_map(d, blockIdx.x, n);
  if (!lm.valid) return;
  const Grp g{lm.sub, d.sub, lm.lead};
  const int k = lm.e, s = lm.s;
  const int e = env_ids ? env_ids[k] : k;
  const int scen = scen_ids[k];
  const pgd_scenario* sc = d.scen + scen;
  const pgd_spawn* sp = d.spawns + (size_t)scen * d.sstride + s;
  MapView mv = map_view(d, sc->map);
  Veh r;
  reset_vehicle(*sp, r, s, s < A);
  RouteCtx ctx;
  if (r.status != ST_EMPTY) after_step_vehicle(mv, g, *sp, r, s < A, true, ctx);
  const unsigned long long am = __ballot(lm.sub == 0 && s < A && r.status == ST_ACTIVE);  // epw == 1 whenever A > 1
  if (s < A && r.status == ST_ACTIVE) r.agent_id = A == 1 ? 0.0f : (float)__popcll(am & ((1ull << lm.lead) - 1ull));
  if (lm.sub != 0) return;
  store_veh(d, e, s, r);
  if (s == 0) {
    d.ei[(size_t)(e) * PGD_NEI + EI_NEXT_AGENT] = A == 1 ? 1 : __popcll(am);
    d.ei[(size_t)(e) * PGD_NEI + EI_AUX] = d.scen[scen].aux;  // parking: free spaces of the new episode
    d.ei[(size_t)(e) * PGD_NEI + EI_SCEN] = scen;
    d.ei[(size_t)(e) * PGD_NEI + EI_NEXT_GROUP] = 0;
    d.ei[(size_t)(e) * PGD_NEI + EI_EP_STEPS] = 0;
    d.ei[(size_t)(e) * PGD_NEI + EI_EPISODES] = 0;
    d.ei[(size_t)(e) * PGD_NEI + EI_STEPS_TOTAL] = 0;
  }
}

// engine.after_step on the current state (used after pgd_set_state)
__global__ __launch_bounds__(WAVE) void k_refresh(PgdDev d) {
  const int V = d.V, A = d.A, N = d.N;
  const LaneMap lm = lane_map(d, blockIdx.x, N);
  if (!lm.valid) return;
  const Grp g{lm.sub, d.sub, lm.lead};
  const int e = lm.e, s = lm.s;
  Veh r;
  load_veh(d, e, s, r);
  if (r.status != ST_ACTIVE && r.status != ST_PENDING && r.status != ST_DYING) return;
  int scen = d.ei[(size_t)(e) * PGD_NEI + EI_SCEN];
  MapView mv = map_view(d, d.scen[scen].map);
  RouteCtx ctx;
  after_step_vehicle(mv, g, d.spawns[(size_t)scen * d.sstride + r.spawn], r, s < A, true, ctx);
  if (lm.sub == 0) store_veh(d, e, s, r);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_observe: stand-alone observation kernel, one block per (env, agent).  pgd_step fuses the observation into k_step when
// a wave carries exactly one env; this kernel serves pgd_reset / pgd_observe and the configurations that do not fuse.
// ---------------------------------------------------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_observe(PgdDev d, float* __restrict__ obs, const uint32_t* __restrict__ flags) {
  __shared__ ObsLds L;
  const int V = d.V, A = d.A, D = d.D;
  const int e = blockIdx.x / A, a = blockIdx.x - e * A;
  const int tid = threadIdx.x;
  const VehRec* recs = d.rec + (size_t)e * V;  // the env's vehicle records
  const VehRec& mine = recs[a];
  float* row = obs + ((size_t)e * A + a) * D;
  // which slots get a row: after a multi-agent step the ones that reported or were (re)spawned, else the active ones
  bool want = mine.i[SI_STATUS] == ST_ACTIVE;
  if (flags) {
    const uint32_t fa = flags[(size_t)e * A + a];
    want = (fa & PGD_F_RESET) ? want : (fa & (PGD_F_REPORT | PGD_F_NEW)) != 0;  // after a reset only the new episode counts
  }
  if (!want) {
    for (int k = tid; k < D; k += BLOCK) row[k] = 0.0f;
    return;
  }
  AgentView ag;
  ag.x = mine.f[SF_X]; ag.y = mine.f[SF_Y]; ag.th = mine.f[SF_THETA];
  sincosf(ag.th, &ag.hy, &ag.hx);
  ag.dl = mine.f[SF_DIST_LEFT]; ag.dr = mine.f[SF_DIST_RIGHT]; ag.v = mine.f[SF_SPEED]; ag.steer = mine.f[SF_STEER];
  ag.a0s = mine.f[SF_ACT0S]; ag.a0t = mine.f[SF_ACT0T]; ag.lhx = mine.f[SF_LASTHX]; ag.lhy = mine.f[SF_LASTHY];
  const int scen = d.ei[(size_t)(e) * PGD_NEI + EI_SCEN];
  const pgd_spawn* spb = d.spawns + (size_t)scen * d.sstride;
  if (tid < WAVE) {  // wave 0: broad phase r = lidar distance (lidar.py:109-124), compacted into LDS
    bool present = false, is_vehicle = true;
    float x = 0, y = 0, ux = 1, uy = 0, hl = 0, hw = 0, spd = 0;
    if (tid < V && d.cfg.num_lasers > 0) {
      int st = recs[tid].i[SI_STATUS];
      present = st == ST_PENDING || st == ST_ACTIVE || st == ST_DYING;
      bool still = st == ST_DYING;  // a finished agent is a static body (zero velocity)
      if (flags && tid < A) {
        // multi-agent step: rows of agents that drove this step show the world before the finishes / respawns
        // (base_env.py:303-344 runs before multi_agent_pgdrive.py:128-141); an agent spawned this step sees the world at
        // its spawn time, i.e. the earlier spawns of the step only
        const uint32_t fa = flags[(size_t)e * A + a], fo = flags[(size_t)e * A + tid];
        if (fa & PGD_F_RESET) {
        } else if (fa & PGD_F_NEW) {
          present = present && (!(fo & PGD_F_NEW) || recs[tid].f[SF_AGENT_ID] < mine.f[SF_AGENT_ID]);
        } else {
          present = (fo & PGD_F_REPORT) || (present && !(fo & PGD_F_NEW));
          still = still && !(fo & PGD_F_REPORT);
        }
      }
      x = recs[tid].f[SF_X]; y = recs[tid].f[SF_Y];
      sincosf(recs[tid].f[SF_THETA], &uy, &ux);
      const pgd_spawn& so = spb[recs[tid].i[SI_SPAWN]];
      hl = 0.5f * so.length; hw = so.kind == PGD_OBJ_CYLINDER ? -1.0f : 0.5f * so.width;
      is_vehicle = so.kind == PGD_OBJ_VEHICLE;
      spd = still ? 0.0f : speed_kmh(recs[tid].f[SF_SPEED]);
    }
    obs_compact<true>(L, tid, a, present, is_vehicle, x, y, ux, uy, hl, hw, spd, ag.x, ag.y, d.cfg.lidar_dist);
  }
  __syncthreads();
  MapView mv = map_view_of(d, d.scen_map + scen);
  const pgd_spawn& msp = spb[mine.i[SI_SPAWN]];
  const RouteCtx ctx = route_ctx(mv, msp, mine.i[SI_CK0], mine.i[SI_CK1]);
  ag.cur_first = ctx.cur_first; ag.cur_n = ctx.cur_n; ag.next_first = ctx.next_first;
  ag.blk = ctx.blk; ag.toll_time = mine.f[SF_PID_HP];
  ag.env = e; ag.slot = a; ag.tick = (uint32_t)d.ei[(size_t)(e) * PGD_NEI + EI_STEPS_TOTAL];
  observe_agent<true>(d, mv, msp, ag, L, row, tid, BLOCK);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
struct pgd_engine {
  PgdDev d;
  int device;
  hipStream_t stream;
  bool own_stream;
  hipEvent_t ev0, ev1;
  bool ev_valid;
  pgd_map* maps; pgd_lane* lanes; pgd_road* roads; pgd_box* boxes; int32_t* cell_start; int32_t* cell_items;
  pgd_box* cell_boxes;
  LaneExt* cell_ext;
  pgd_map* scen_map;  // per scenario: copy of its map header (saves one dependent load per block)
  std::vector<pgd_map>* h_maps;
  std::vector<pgd_scenario>* h_scen;
  pgd_scenario* scen; pgd_spawn* spawns;
  int32_t* d_ids;  // scratch [2N]
  bool have_maps, have_scen;
  // per-kernel HIP-event profile of pgd_step launches (bench.py roofline numbers)
  std::vector<hipEvent_t>* prof_ev;  // 3 events per recorded step
  int prof_cap, prof_n, prof_stride, prof_tick;
  bool prof_grouped;
  bool has_objects;  // some spawn record is a traffic object (pgd_upload_scenarios): selects the OBJ kernels
  bool step_timing;  // record ev0 / ev1 around every step (pgd_last_step_ms)
  bool no_fuse;      // PGD_NO_FUSE was set when the engine was created (debug: always run the stand-alone k_observe)
  bool prof_fused;
};

template <typename T>
static int upload(T** dst, const T* src, size_t n, hipStream_t st) {
  if (*dst) { HIPCHK(hipFree(*dst)); *dst = nullptr; }
  HIPCHK(hipMalloc(dst, sizeof(T) * (n ? n : 1)));
  if (n) HIPCHK(hipMemcpyAsync(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  return PGD_OK;
}

static int build_scen_map(pgd_engine* h) {
  if (!h->h_maps || !h->h_scen) return PGD_OK;
  std::vector<pgd_map> sm(h->h_scen->size());
  for (size_t k = 0; k < sm.size(); ++k) {
    int m = (*h->h_scen)[k].map;
    if (m < 0 || m >= (int)h->h_maps->size()) return PGD_ERR_ARG;
    sm[k] = (*h->h_maps)[m];
  }
  int rc = upload(&h->scen_map, sm.data(), sm.size(), h->stream);
  if (rc) return rc;
  h->d.scen_map = h->scen_map;
  return PGD_OK;
}

extern "C" {

const char* pgd_version(void) { return "pgdrive_hip 0.1 (gfx950)"; }

int pgd_obs_dim(const pgd_config* c) {
  const int toll = (c->marl_flags & PGD_MA_TOLLGATE) != 0;
  return (c->side_lasers > 0 ? c->side_lasers : 2) + 6 + c->lane_line_lasers + (c->random_agent_model ? 2 : 0) +
         (toll ? 0 : PGD_NAVI_DIM) + 4 * c->num_others + c->num_lasers + (toll ? 2 : 0);
}

int pgd_create(const pgd_config* cfg, int device, void* hip_stream, pgd_handle* out) {
  if (!cfg || !out) return PGD_ERR_ARG;
  int V = cfg->num_agents + cfg->num_traffic;
  if (cfg->num_envs <= 0 || cfg->num_agents <= 0 || V > MAXV || cfg->num_others > 16 || cfg->num_lasers < 0) return PGD_ERR_ARG;
  HIPCHK(hipSetDevice(device));
  pgd_engine* h = (pgd_engine*)calloc(1, sizeof(pgd_engine));
  h->device = device;
  h->no_fuse = getenv("PGD_NO_FUSE") != nullptr;
  h->d.cfg = *cfg;
  h->d.N = cfg->num_envs; h->d.A = cfg->num_agents; h->d.T = cfg->num_traffic; h->d.V = V;
  h->d.D = pgd_obs_dim(cfg);
  h->d.NV = h->d.N * V;
  const bool marl = (cfg->marl_flags & PGD_MA_ENABLED) != 0;
  // multi-agent engines have no IDM traffic; num_traffic slots may hold static bodies (toll booths, group PGD_GROUP_NEVER)
  if (marl && (cfg->respawn_places < 0 || cfg->respawn_dests < 0)) return PGD_ERR_ARG;
  h->d.sstride = V + (marl ? cfg->respawn_places * cfg->respawn_dests : 0);
  h->d.sub = WAVE / V < 16 ? WAVE / V : 16;  // sub-lanes per vehicle
  h->d.epw = WAVE / (V * h->d.sub);          // whole environments per wave
  if (hip_stream) { h->stream = (hipStream_t)hip_stream; h->own_stream = false; }
  else { HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
  HIPCHK(hipEventCreate(&h->ev0));
  HIPCHK(hipEventCreate(&h->ev1));
  size_t nv = (size_t)h->d.NV;
  HIPCHK(hipMalloc(&h->d.rec, sizeof(VehRec) * nv));
  HIPCHK(hipMalloc(&h->d.ei, sizeof(int32_t) * (size_t)h->d.N * PGD_NEI));
  HIPCHK(hipMalloc(&h->d_ids, sizeof(int32_t) * (size_t)h->d.N * 2));
  HIPCHK(hipMemsetAsync(h->d.rec, 0, sizeof(VehRec) * nv, h->stream));
  HIPCHK(hipMemsetAsync(h->d.ei, 0, sizeof(int32_t) * (size_t)h->d.N * PGD_NEI, h->stream));
  *out = h;
  return PGD_OK;
}

int pgd_upload_maps(pgd_handle h, const pgd_map* maps, int n_maps, const pgd_lane* lanes, int n_lanes,
                    const pgd_road* roads, int n_roads, const pgd_box* boxes, int n_boxes, const int32_t* cs, int n_cs,
                    const int32_t* ci, int n_ci) {
  if (!h || !maps || n_maps <= 0) return PGD_ERR_ARG;
  HIPCHK(hipSetDevice(h->device));
  int rc;
  if ((rc = upload(&h->maps, maps, n_maps, h->stream))) return rc;
  {
    // device copy of the lane table: `pad` carries the lane count of the lane's road (the lanes of a road are consecutive,
    // so neighbour lanes follow from the lane id alone: no road-record read in the IDM neighbour search)
    std::vector<pgd_lane> dl(lanes, lanes + n_lanes);
    for (int m = 0; m < n_maps; ++m)
      for (int k = 0; k < maps[m].n_lanes; ++k) {
        pgd_lane& L = dl[(size_t)maps[m].lane_off + k];
        if (L.road < 0 || L.road >= maps[m].n_roads) return PGD_ERR_ARG;
        const pgd_road& R = roads[maps[m].road_off + L.road];
        if (R.first_lane + L.index != k) return PGD_ERR_ARG;
        L.pad = R.n_lanes;
      }
    if ((rc = upload(&h->lanes, dl.data(), n_lanes, h->stream))) return rc;
  }
  if ((rc = upload(&h->roads, roads, n_roads, h->stream))) return rc;
  if ((rc = upload(&h->boxes, boxes, n_boxes, h->stream))) return rc;
  if ((rc = upload(&h->cell_items, ci, n_ci, h->stream))) return rc;
  // cell-major copies of the boxes: cell_boxes[item_off + k] = boxes[box_off + cell_items[item_off + k]] — one dependent
  // load less per box in the grid walks, and a cell's boxes are contiguous (coalesced across sub-lanes)
  {
    // Inside each cell the lane-surface boxes are moved to the front (stable), and the device copy of cell_start packs
    // the number of lane boxes into the top byte (see cell_first / cell_mid).
    std::vector<pgd_box> cb((size_t)(n_ci > 0 ? n_ci : 1));
    std::vector<LaneExt> cx((size_t)(n_ci > 0 ? n_ci : 1), LaneExt{0.f, 0.f, 0.f, -1});
    std::vector<int32_t> cs2(cs, cs + n_cs);
    int max_lanes = 0, max_roads = 0;
    for (int m = 0; m < n_maps; ++m) {
      const pgd_map& M = maps[m];
      const int n_cells = M.gx * M.gy;
      if (cs[M.cell_off + n_cells] >= (1 << 24)) return PGD_ERR_ARG;
      for (int c = 0; c < n_cells; ++c) {
        const int a = cs[M.cell_off + c], b = cs[M.cell_off + c + 1];
        int w = a;
        for (int pass = 0; pass < 2; ++pass)
          for (int k = a; k < b; ++k) {
            const pgd_box& bx = boxes[M.box_off + ci[M.item_off + k]];
            if ((bx.kind == PGD_BOX_LANE) == (pass == 0)) {
              if (pass == 0) {
                if (bx.lane < 0 || bx.lane >= M.n_lanes) return PGD_ERR_ARG;
                const pgd_lane& L = lanes[M.lane_off + bx.lane];
                cx[(size_t)M.item_off + w] = L.dir == 0.0f ? LaneExt{L.bx, L.by, 0.0f, L.road} : LaneExt{L.ax, L.ay, L.dir, L.road};
              }
              cb[(size_t)M.item_off + w++] = bx;
            }
            if (pass == 0 && k == b - 1) {
              const int n_lane_boxes = w - a;
              if (n_lane_boxes > 255) return PGD_ERR_ARG;
              cs2[M.cell_off + c] = a | (n_lane_boxes << 24);
            }
          }
      }
      if (M.n_lanes > max_lanes) max_lanes = M.n_lanes;
      if (M.n_roads > max_roads) max_roads = M.n_roads;
    }
    if ((rc = upload(&h->cell_boxes, cb.data(), (size_t)n_ci, h->stream))) return rc;
    if ((rc = upload(&h->cell_ext, cx.data(), (size_t)n_ci, h->stream))) return rc;
    if ((rc = upload(&h->cell_start, cs2.data(), n_cs, h->stream))) return rc;
    size_t need = (size_t)max_lanes * sizeof(pgd_lane) + (size_t)max_roads * sizeof(pgd_road);
    h->d.lds_bytes = (h->d.epw == 1 && need <= 40 * 1024) ? (int)need : 0;  // else: tables stay in global memory
    // Measured (profiles/r01_notes.md): staging costs a 36 MB L2 burst per step and loses to plain global reads once the
    // IDM search stopped touching the lane table (58 vs 49 M env-steps/s); kept as an opt-in for larger-V experiments.
    if (!getenv("PGD_LDS_TABLES")) h->d.lds_bytes = 0;
  }
  h->d.maps = h->maps; h->d.lanes = h->lanes; h->d.roads = h->roads; h->d.boxes = h->boxes;
  h->d.cell_start = h->cell_start; h->d.cell_items = h->cell_items; h->d.cell_boxes = h->cell_boxes; h->d.cell_ext = h->cell_ext;
  if (!h->h_maps) h->h_maps = new std::vector<pgd_map>();
  h->h_maps->assign(maps, maps + n_maps);
  if ((rc = build_scen_map(h))) return rc;
  h->have_maps = true;
  return PGD_OK;
}

int pgd_upload_scenarios(pgd_handle h, const pgd_scenario* scen, int n_scen, const pgd_spawn* spawns) {
  if (!h || !scen || n_scen <= 0 || !spawns) return PGD_ERR_ARG;
  HIPCHK(hipSetDevice(h->device));
  int rc;
  if ((rc = upload(&h->scen, scen, n_scen, h->stream))) return rc;
  if ((rc = upload(&h->spawns, spawns, (size_t)n_scen * h->d.sstride, h->stream))) return rc;
  h->d.scen = h->scen; h->d.spawns = h->spawns; h->d.n_scen = n_scen;
  h->has_objects = false;
  for (size_t k = 0; k < (size_t)n_scen * h->d.sstride; ++k)
    if (spawns[k].lane >= 0 && spawns[k].kind != PGD_OBJ_VEHICLE) { h->has_objects = true; break; }
  if (!h->h_scen) h->h_scen = new std::vector<pgd_scenario>();
  h->h_scen->assign(scen, scen + n_scen);
  if ((rc = build_scen_map(h))) return rc;
  h->have_scen = true;
  return PGD_OK;
}

static int launch_observe(pgd_handle h, float* d_obs, const uint32_t* d_flags) {
  int blocks = h->d.N * h->d.A;
  if (h->d.cfg.num_lasers > 128)  // up to 128 beams one wave does it in two rounds: 4x fewer waves than 256-thread blocks
    hipLaunchKernelGGL(k_observe<256>, dim3(blocks), dim3(256), 0, h->stream, h->d, d_obs, d_flags);
  else hipLaunchKernelGGL(k_observe<64>, dim3(blocks), dim3(64), 0, h->stream, h->d, d_obs, d_flags);
  HIPCHK(hipGetLastError());
  return PGD_OK;
}

int pgd_reset(pgd_handle h, const int32_t* env_ids, const int32_t* scen_ids, int n, float* d_obs) {
  if (!h || !scen_ids || n <= 0 || n > h->d.N) return PGD_ERR_ARG;
  if (!h->have_maps || !h->have_scen) return PGD_ERR_STATE;
  HIPCHK(hipSetDevice(h->device));
  for (int k = 0; k < n; ++k) {
    if (scen_ids[k] < 0 || scen_ids[k] >= h->d.n_scen) return PGD_ERR_ARG;
    if (env_ids && (env_ids[k] < 0 || env_ids[k] >= h->d.N)) return PGD_ERR_ARG;
  }
  int32_t* d_env = nullptr;
  if (env_ids) {
    HIPCHK(hipMemcpyAsync(h->d_ids, env_ids, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->stream));
    d_env = h->d_ids;
  }
  HIPCHK(hipMemcpyAsync(h->d_ids + h->d.N, scen_ids, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->stream));
  int blocks = (n + h->d.epw - 1) / h->d.epw;
  hipLaunchKernelGGL(k_reset, dim3(blocks), dim3(WAVE), 0, h->stream, h->d, d_env, h->d_ids + h->d.N, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));  // host id buffers may be reused by the caller
  if (d_obs) return launch_observe(h, d_obs, nullptr);
  return PGD_OK;
}

int pgd_step(pgd_handle h, const float* d_actions, float* d_obs, float* d_reward, uint8_t* d_done, uint32_t* d_flags) {
  if (!h || !d_actions || !d_reward || !d_done || !d_flags) return PGD_ERR_ARG;
  if (!h->have_maps || !h->have_scen) return PGD_ERR_STATE;
  const bool marl = (h->d.cfg.marl_flags & PGD_MA_ENABLED) != 0;
  const bool fuse = d_obs && !marl && h->d.epw == 1 && h->d.A <= FUSE_MAX_AGENTS && !h->no_fuse;
  bool prof = h->prof_ev && h->prof_n < h->prof_cap;
  // strided profile: with the observation fused (one kernel per step) events [0] / [1] bracket a GROUP of `stride`
  // back-to-back launches and the group time is divided by the stride; otherwise every stride-th step is bracketed
  const bool grouped = prof && h->prof_stride > 1 && fuse;
  bool g_open = false, g_close = false;
  if (prof && h->prof_stride > 1) {
    const int ph = h->prof_tick++ % h->prof_stride;
    if (grouped) { g_open = ph == 0; g_close = ph == h->prof_stride - 1; prof = false; }
    else prof = ph == 0;
  }
  hipEvent_t* pe = (prof || g_open || g_close) ? &(*h->prof_ev)[(size_t)h->prof_n * 3] : nullptr;
  const bool timing = h->step_timing && !prof && !grouped;
  if (prof || timing || g_open) HIPCHK(hipEventRecord((prof || g_open) ? pe[0] : h->ev0, h->stream));
  int blocks = (h->d.N + h->d.epw - 1) / h->d.epw;
  if (marl && h->d.epw != 1) return PGD_ERR_STATE;  // the multi-agent tail needs the env in one wave (V >= 33 or SUB split)
  void (*kern)(PgdDev, const float*, float*, uint8_t*, uint32_t*, float*) = k_step<false, false, false>;
  if (marl) kern = h->has_objects ? k_step<true, true, true> : k_step<true, true, false>;  // objects = toll booths
  else if (h->d.epw == 1) kern = h->has_objects ? k_step<true, false, true> : k_step<true, false, false>;
  else if (h->has_objects) kern = k_step<false, false, true>;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVE), (size_t)h->d.lds_bytes, h->stream, h->d, d_actions, d_reward, d_done,
                     d_flags, fuse ? d_obs : (float*)nullptr);
  HIPCHK(hipGetLastError());
  if (prof || g_close) HIPCHK(hipEventRecord(pe[1], h->stream));
  if (g_close) h->prof_n += 1;
  if (d_obs && !fuse) {
    int rc = launch_observe(h, d_obs, marl ? d_flags : (const uint32_t*)nullptr);
    if (rc) return rc;
  }
  h->prof_fused = fuse;
  h->prof_grouped = grouped;
  if ((prof && !fuse) || timing) HIPCHK(hipEventRecord(prof ? pe[2] : h->ev1, h->stream));  // fused: [0],[1] bracket the only kernel
  if (prof) h->prof_n += 1;
  else if (timing) h->ev_valid = true;
  return PGD_OK;
}

int pgd_observe(pgd_handle h, float* d_obs) {
  if (!h || !d_obs) return PGD_ERR_ARG;
  if (!h->have_maps || !h->have_scen) return PGD_ERR_STATE;
  int blocks = (h->d.N + h->d.epw - 1) / h->d.epw;
  hipLaunchKernelGGL(k_refresh, dim3(blocks), dim3(WAVE), 0, h->stream, h->d);
  HIPCHK(hipGetLastError());
  return launch_observe(h, d_obs, nullptr);
}

int pgd_state_dims(pgd_handle h, int* nf, int* ni, int* nei) {
  (void)h;
  if (nf) *nf = PGD_NF;
  if (ni) *ni = PGD_NI;
  if (nei) *nei = PGD_NEI;
  return PGD_OK;
}
}  // extern "C" (state conversion helpers are C++ templates)

// ABI order is field-major ([field][env*V + slot], [field][env]); the device keeps one 128 B record per vehicle and one
// PGD_NEI-int row per env — converted on the host
extern "C" int pgd_get_state(pgd_handle h, float* f, int32_t* i, int32_t* ei) {
  if (!h || !f || !i || !ei) return PGD_ERR_ARG;
  const size_t nv = (size_t)h->d.NV;
  const int N = h->d.N;
  std::vector<VehRec> tr(nv);
  std::vector<int32_t> te((size_t)N * PGD_NEI);
  HIPCHK(hipMemcpyAsync(tr.data(), h->d.rec, sizeof(VehRec) * nv, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(te.data(), h->d.ei, sizeof(int32_t) * te.size(), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (size_t k = 0; k < nv; ++k) {
    for (int q = 0; q < PGD_NF; ++q) f[(size_t)q * nv + k] = tr[k].f[q];
    for (int q = 0; q < PGD_NI; ++q) i[(size_t)q * nv + k] = tr[k].i[q];
  }
  for (int e = 0; e < N; ++e)
    for (int q = 0; q < PGD_NEI; ++q) ei[(size_t)q * N + e] = te[(size_t)e * PGD_NEI + q];
  return PGD_OK;
}
extern "C" int pgd_set_state(pgd_handle h, const float* f, const int32_t* i, const int32_t* ei) {
  if (!h || !f || !i || !ei) return PGD_ERR_ARG;
  const size_t nv = (size_t)h->d.NV;
  const int N = h->d.N;
  std::vector<VehRec> tr(nv);
  std::vector<int32_t> te((size_t)N * PGD_NEI);
  for (size_t k = 0; k < nv; ++k) {
    for (int q = 0; q < PGD_NF; ++q) tr[k].f[q] = f[(size_t)q * nv + k];
    for (int q = 0; q < PGD_NI; ++q) tr[k].i[q] = i[(size_t)q * nv + k];
  }
  for (int e = 0; e < N; ++e)
    for (int q = 0; q < PGD_NEI; ++q) te[(size_t)e * PGD_NEI + q] = ei[(size_t)q * N + e];
  HIPCHK(hipMemcpyAsync(h->d.rec, tr.data(), sizeof(VehRec) * nv, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d.ei, te.data(), sizeof(int32_t) * te.size(), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return PGD_OK;
}

extern "C" {
int pgd_enable_step_timing(pgd_handle h, int on) {
  if (!h) return PGD_ERR_ARG;
  h->step_timing = on != 0;
  h->ev_valid = false;
  return PGD_OK;
}

int pgd_last_step_ms(pgd_handle h, float* ms) {
  if (!h || !ms) return PGD_ERR_ARG;
  if (!h->ev_valid) return PGD_ERR_STATE;  // pgd_enable_step_timing(h, 1) and a step first
  HIPCHK(hipEventSynchronize(h->ev1));
  HIPCHK(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return PGD_OK;
}

int pgd_profile_begin(pgd_handle h, int capacity) { return pgd_profile_begin_strided(h, capacity, 1); }

int pgd_profile_begin_strided(pgd_handle h, int capacity, int stride) {
  if (!h || capacity <= 0 || stride <= 0) return PGD_ERR_ARG;
  h->prof_stride = stride;
  h->prof_tick = 0;
  if (!h->prof_ev) h->prof_ev = new std::vector<hipEvent_t>();
  while ((int)h->prof_ev->size() < capacity * 3) {
    hipEvent_t ev;
    HIPCHK(hipEventCreate(&ev));
    h->prof_ev->push_back(ev);
  }
  h->prof_cap = capacity;
  h->prof_n = 0;
  return PGD_OK;
}

int pgd_profile_end(pgd_handle h, float* k_step_ms, float* k_observe_ms, int* count) {
  if (!h || !h->prof_ev || !k_step_ms || !k_observe_ms || !count) return PGD_ERR_ARG;
  HIPCHK(hipStreamSynchronize(h->stream));
  double a = 0.0, b = 0.0;
  for (int k = 0; k < h->prof_n; ++k) {
    float t0 = 0.f, t1 = 0.f;
    HIPCHK(hipEventElapsedTime(&t0, (*h->prof_ev)[(size_t)k * 3], (*h->prof_ev)[(size_t)k * 3 + 1]));
    if (!h->prof_fused) HIPCHK(hipEventElapsedTime(&t1, (*h->prof_ev)[(size_t)k * 3 + 1], (*h->prof_ev)[(size_t)k * 3 + 2]));
    a += h->prof_grouped ? t0 / (float)h->prof_stride : t0;
    b += t1;
  }
  *count = h->prof_n;
  *k_step_ms = h->prof_n ? (float)(a / h->prof_n) : 0.f;
  *k_observe_ms = h->prof_n ? (float)(b / h->prof_n) : 0.f;
  h->prof_cap = 0;
  h->prof_n = 0;
  return PGD_OK;
}

#ifdef PGD_PROF
int pgd_debug_phase_cycles(pgd_handle h, unsigned long long* out64, int reset) {
  HIPCHK(hipStreamSynchronize(h->stream));
  std::vector<unsigned long long> all((size_t)PROF_BLOCKS * 32);
  HIPCHK(hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(g_phase_cycles), sizeof(unsigned long long) * all.size()));
  for (int k = 0; k < 64; ++k) out64[k] = 0;  // [0,32): sums over blocks, [32,64): max over blocks
  for (size_t b = 0; b < PROF_BLOCKS; ++b)
    for (int k = 0; k < 32; ++k) {
      out64[k] += all[b * 32 + k];
      if (all[b * 32 + k] > out64[32 + k]) out64[32 + k] = all[b * 32 + k];
    }
  if (reset) {
    std::fill(all.begin(), all.end(), 0ull);
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), all.data(), sizeof(unsigned long long) * all.size()));
  }
  return PGD_OK;
}
#endif

int pgd_sync(pgd_handle h) {
  if (!h) return PGD_ERR_ARG;
  HIPCHK(hipStreamSynchronize(h->stream));
  return PGD_OK;
}

int pgd_destroy(pgd_handle h) {
  if (!h) return PGD_ERR_ARG;
  (void)hipStreamSynchronize(h->stream);
  void* bufs[] = {h->d.rec, h->d.ei, h->d_ids, h->maps, h->lanes, h->roads, h->boxes, h->cell_start,
                  h->cell_items, h->cell_boxes, h->cell_ext, h->scen_map, h->scen, h->spawns};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  (void)hipEventDestroy(h->ev0);
  (void)hipEventDestroy(h->ev1);
  if (h->prof_ev) {
    for (hipEvent_t ev : *h->prof_ev) (void)hipEventDestroy(ev);
    delete h->prof_ev;
  }
  delete h->h_maps;
  delete h->h_scen;
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  free(h);
  return PGD_OK;
}

}  // extern "C"
