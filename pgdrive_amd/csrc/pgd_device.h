// pgd_device.h — fp32 device routines of the batched PGDrive step engine (gfx950 / CDNA4, wave64).
//
// Every routine cites the reference code whose behaviour it reproduces (paths relative to pgdrive/ in
// decisionforce/pgdrive v0.1.4).  Bullet queries of the reference are evaluated as exact 2-D geometry on the boxes the
// reference registers in Bullet (uploaded by pgd_upload_maps); see DESIGN.md for the data layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pgd_state_layout.h"
#include "../../include/pgdrive_hip.h"

#define PGD_PI 3.14159265358979323846f
#define DEV __device__ __forceinline__
#define DEV_HOST __host__ __device__ inline

// One 16-byte piece of a vehicle record in memory (struct Veh below = eight pieces).  A BLOCK of n records -- the V slots of an env,
// the slots of a scenario's reset image, a scenario's respawn records -- is stored as eight PLANES of n pieces: piece k of record s
// lies (k * n + s) * 16 bytes behind the block's start (rec_block / load_rec / store_rec in pgd_vehicle.h).  The lanes of a wave hold
// one vehicle each and read their records piece by piece: with a line per vehicle every one of those eight instructions touched as
// many cache lines as the env has vehicles (17 ... 40), plane by plane it touches n / 8 of them.  Same bytes, same DRAM pages;
// `tools/rec_layout.hip`: the 40 records of 4096 envs read and written back in 9.6 instead of 13.3 us, 17 records read 0.4 us sooner.
// (An opaque type: nothing indexes a record as an array element any more.)
struct RecPiece { uint4 q; };

// ---------------------------------------------------------------------------------------------------------------------
// device-side view of the engine
// ---------------------------------------------------------------------------------------------------------------------
struct PgdDev {
  pgd_config cfg;
  int N, A, T, V, D, NV;
  int epw;        // whole environments per wave in k_step
  int sub;        // sub-lanes cooperating on one vehicle
  int pack_obs;   // several envs per wave (throughput mode): k_step appends the lidar observation of every env of the wave
  int sstride;    // pgd_spawn records per scenario: V slots + respawn_places * respawn_dests (multi-agent)
  const pgd_map* maps;
  const pgd_lane* lanes;
  const pgd_road* roads;
  const pgd_box* boxes;
  const int32_t* cell_start;
  const int32_t* cell_items;
  const pgd_box* cell_boxes;  // cell-major copies of the boxes, lane boxes first inside each cell (pgd_upload_maps)
  const struct LaneExt* cell_ext;  // same indexing: what localisation needs from the lane of a lane box
  const struct LaneNav* lane_nav;  // per lane: what the navigation block of the observation needs
  const pgd_scenario* scen;
  const pgd_map* scen_map;    // [n_scen] copy of each scenario's map header
  const pgd_spawn* spawns;
  const float2* spawn_hv;     // [n_scen * sstride] (cos, sin) of each spawn heading (k_spawn_hv at upload)
  int n_scen;
  RecPiece* rec;  // [N] blocks of V records of 128 bytes, piece planes (above; device layout, the ABI blobs are field-major)
  int32_t* ei;         // [N][PGD_NEI]
  pgd_map* env_map;    // [N] copy of the map header of the env's running scenario (rewritten on reset): the map view of a
                       // step needs no env -> scenario -> header chain
  int use_imask;       // 0: every slot is read from the env's own record (one env per wave: one dependent load level less)
  int no_groups;       // no uploaded scenario has a traffic trigger group (pgd_upload_scenarios): the trigger test of a step is skipped
  unsigned long long* imask;  // [N] bit s: slot s of the env still equals its scenario's reset image (never stored since)
  const RecPiece* reset_img;  // [n_scen] blocks of V records: every slot right after a reset of its scenario (k_reset_image)
  const RecPiece* respawn_img;  // [n_scen] blocks of sstride - V records (multi-agent): an agent right after it was spawned from respawn record V + k
  const float2* beam;  // [num_lasers] (cos, sin) of the beam angle i * 2 pi / num_lasers in the vehicle frame
  // output addressing of one launch: the observation row of (env e, agent a) starts at obs + e * ostride + a * D.
  // pgd_step: ostride = A * D (dense [N, A, D]).  pgd_step_packed: ostride = the caller's row stride and `prow` = the same
  // buffer: reward and done (as 0 / 1 floats) are written behind the A * D observation floats of the env's row, so the row
  // is the unit of the per-step gather (pgdrive_hip.h) and no copy kernel packs it
  int ostride;
  float* prow;
  int dbg_exit;  // exit-profile builds only (PGD_EXITAT)
  int unit_off;  // first block unit of this launch (pgd_step_group); 0 for a whole-engine step
  int obs_g;     // multi-agent k_step with the fused observation: observers per pass (what fits the step's LDS)
  // multi-agent engines: rowz[e] = (mask, tag).  Bit a of the mask = row a of env e in the observation buffer that `tag` names (its
  // address mixed with the row stride, rowz_tag) was written as zeros by an earlier call and has not been due since: not written
  // again (40 agent slots, 6 - 30 alive: the zero rows were 46 MB of stores per step at 4096 envs).  The buffer's identity lives HERE,
  // per env, and is compared by the kernel itself: a HIP graph replayed after an eager call with another buffer, or env groups on
  // their own streams, cannot get it wrong (round 4 kept it in a host-side cache: ADVICE r04).  Null (PGD_NO_ROWZ=1): every row that
  // is not due is zero-filled by every call -- for callers that post-process the returned rows in place.
  ulonglong2* rowz;
  uint8_t* bev_fill;  // [N] or null: the env was reset -- the top-down observation refills its history (pgd_topdown.h)
  // multi-agent engines whose rows are written by k_observe_env after the step (many agent slots): the observation buffer of this
  // step, or null.  k_step then writes the STATE BLOCK of every row that is due itself -- one lane per agent, from the record, the
  // route context and the map view it holds in registers -- and k_observe_env<..., STATE = false> does the pairwise part only
  // (round 5: the state phase of the four-wave kernel was 2.8 k of its 7.4 k instructions per env, every wave running the whole
  // branch ladder for six lanes per agent after reading records, spawn records and lane tables back from memory).
  float* state_rows;
  // Every body of every uploaded scenario is a vehicle box of ONE size (the multi-agent defaults: one vehicle model, no traffic
  // objects; pgd_upload_scenarios compares the spawn records): its length and width, else 0.  The multi-agent observation kernel then
  // publishes the bodies without the spawn record of each -- a second memory round trip behind the records (round 6).
  float uni_len, uni_wid;
};

// Device-side vehicle record = the per-lane register image of a vehicle (device-private; pgd_get_state / pgd_set_state
// convert to the ABI's field-major blobs).  128 B = eight 16-byte pieces: a lane loads / stores its vehicle with 8 dwordx4
// transactions and no field shuffling -- the struct in registers IS the record; in memory its pieces lie in the planes of its
// block (RecPiece above).
// Besides the ABI fields of include/pgd_state_layout.h (SF_THROTTLE is not stored: the last applied throttle always equals
// the newer entry of the action deque, SF_ACT1T; base_vehicle.py:343-349 sets both from the same action) the line carries
// state DERIVED from them, kept from step to step instead of being recomputed through dependent table reads every step:
//   hx, hy     unit heading vector (cos, sin of th) -- no sincosf per vehicle and step
//   lon        longitudinal coordinate of the vehicle on its own lane (what the IDM neighbour search reads)
//   road_cur.. route context (Navigation.current_ref_lanes / next_ref_lanes, navigation.py:155-183): road ids, first lanes
//              and lane counts of the current and the next checkpoint pair, block id of the current road; changes only
//              when a checkpoint is passed (route_refresh)
// The int fields are bit-fields: 15 ints live in 6 registers and the compiler extracts them with v_bfe where they are used.
struct __attribute__((aligned(16))) Veh {
  float x, y, th, v;                  // position, heading_theta [rad], speed [m/s]
  float hx, hy, lon, steer;
  uint32_t lane : 16, spawn : 16;
  int32_t rlane : 16;                 // traffic: routing target lane (-1 = None); agents: episode length
  uint32_t timer : 16;                // saturating
  uint32_t vflags : 16, status : 4, ck0 : 6, ck1 : 6;
  uint32_t road_cur : 12, road_next : 12, blk : 8;  // road ids (0xfff = none), Road.block_ID char
  uint32_t cur_first : 16, next_first : 16;
  uint32_t cur_n : 8, next_n : 8, spare : 16;
  float target, agent_id;
  float php, phi, plp, pli;           // IDM PIDs; agents: toll / parking bookkeeping (pgd_state_layout.h)
  float lastx, lasty, lasthx, lasthy;
  float a0s, a0t, a1s, a1t;
  float energy, dl, dr, eprew;
};
typedef Veh VehRec;
static_assert(sizeof(Veh) == 128, "vehicle record must be exactly one 128-byte line");

// Per lane-box extract of its lane (device-private, built on upload): the heading test and the road preference of
// ray_localization (scene_utils.py:158-172, navigation.py:328-344) then need no dependent read of the 64-byte lane record.
struct __attribute__((aligned(16))) LaneExt {
  float ax, ay;  // straight lane: unit direction; arc: centre
  float dir;     // 0 = straight, +-1 = CircularLane.direction
  int32_t road;  // map-local road id of the lane
};

// Per lane, device-private, built on upload: what the navigation block of the observation needs from a reference lane
// (Navigation._get_info_for_checkpoint, navigation.py:213-260).  The check point lane.position(length, lateral) is
// end + lateral * normal for both lane types, so the kernel evaluates no sincos and reads 32 instead of 64 bytes.
struct __attribute__((aligned(16))) LaneNav {
  float ex, ey;   // lane.position(length, 0)
  float nx, ny;   // d position / d lateral at the lane end
  float radius;   // CircularLane.radius (0 on straight lanes)
  float dir;      // 0 = straight, +-1 = CircularLane.direction
  float angle;    // end_phase - start_phase (dir = 1) or start_phase - end_phase (dir = -1), radians
  float pad;
};

// The hot tables (lanes, grid cells and their boxes) are carried as pointers; the road table and the per-lane navigation extract
// are touched once or twice per step and are addressed from the engine view where they are used (`roads()`, `lnav()`): every
// pointer carried through the whole kernel costs two SGPRs, and k_step spills SGPRs.
// the scalars of the map header the step uses after its first store (grid geometry, lane width): read once with the header, at the
// start of the kernel -- a read of the header later on is a vector-memory read (the compiler may not assume the table unchanged
// once the kernel has stored anything), waited for on the spot
struct MapGrid {
  int gx, gy;
  float ox, oy, cell, lane_width;
};
struct MapView {
  const PgdDev* dv;
  const pgd_map* m;
  // the header's scalars, read in place where they are used (the general kernels have no scalar registers left to hold them)
  DEV int gx() const { return m->gx; }
  DEV int gy() const { return m->gy; }
  DEV float ox() const { return m->ox; }
  DEV float oy() const { return m->oy; }
  DEV float cell() const { return m->cell; }
  DEV float lane_width() const { return m->lane_width; }
  const pgd_lane* lanes;
  const int32_t* cstart;
  const LaneExt* cext;
  const pgd_box* cbox;
  int road_off, lane_off;  // (two SGPRs instead of four: the bases come from the kernel arguments at the point of use)
  DEV const pgd_road* roads() const { return dv->roads + road_off; }
  DEV const LaneNav* lnav() const { return dv->lane_nav + lane_off; }
};

DEV MapView map_view_of(const PgdDev& d, const pgd_map* m) {
  MapView v;
  v.dv = &d;
  v.m = m;
  v.lanes = d.lanes + m->lane_off;
  v.road_off = m->road_off; v.lane_off = m->lane_off;
  v.cstart = d.cell_start + m->cell_off;
  v.cext = d.cell_ext + m->item_off;
  v.cbox = d.cell_boxes + m->item_off;
  return v;
}
// the same view with the header's scalars read ahead, with the header itself (kernels specialised for a default configuration:
// their configuration constants leave scalar registers free).  Routines that read those scalars take the view type as a template
// parameter; everything else takes the base.
struct MapViewPre : MapView {
  MapGrid grid;
  DEV int gx() const { return grid.gx; }
  DEV int gy() const { return grid.gy; }
  DEV float ox() const { return grid.ox; }
  DEV float oy() const { return grid.oy; }
  DEV float cell() const { return grid.cell; }
  DEV float lane_width() const { return grid.lane_width; }
};
template <class MV> DEV MV map_view_as(const PgdDev& d, const pgd_map* m);
template <> DEV MapView map_view_as<MapView>(const PgdDev& d, const pgd_map* m) { return map_view_of(d, m); }
template <> DEV MapViewPre map_view_as<MapViewPre>(const PgdDev& d, const pgd_map* m) {
  MapViewPre v;
  static_cast<MapView&>(v) = map_view_of(d, m);
  v.grid = MapGrid{m->gx, m->gy, m->ox, m->oy, m->cell, m->lane_width};
  return v;
}
DEV MapView map_view(const PgdDev& d, int map) { return map_view_of(d, d.maps + map); }

// ---------------------------------------------------------------------------------------------------------------------
// scalar helpers: utils/math_utils.py:32-94, cutils.pyx:147-154
// ---------------------------------------------------------------------------------------------------------------------
DEV float clipf(float a, float lo, float hi) { return fminf(fmaxf(a, lo), hi); }
DEV float norm2(float x, float y) { return sqrtf(x * x + y * y); }
DEV float wrap_to_pi(float x) {  // ((x + pi) % (2 pi)) - pi with Python's sign-of-divisor modulo: result in [-pi, pi)
#pragma clang fp contract(off)
#pragma clang fp reassociate(off)
  float y = x + PGD_PI;
  float r = y - 2.0f * PGD_PI * floorf(y * (0.5f / PGD_PI));
  r = r < 0.0f ? r + 2.0f * PGD_PI : (r >= 2.0f * PGD_PI ? r - 2.0f * PGD_PI : r);
  return r - PGD_PI;
}
// BaseVehicle.heading_theta (base_vehicle.py:411-416): (-getH() - 90) deg with getH in (-180, 180]: [-3 pi / 2, pi / 2)
// (Not pinned like wrap_to_pi itself: with contraction / re-association switched off HERE the metric's kernel ran 0.1 us slower --
// 17.00 -> 17.12 us, three interleaved repetitions -- for a different schedule of the same arithmetic.  Left as it is, SF_THETA of the
// throughput-mode and the one-env kernels differs by one ulp in 40 - 60 of 6.3 M vehicle-steps (`(th + pi / 2) + pi` folded in one of
// them, tools/mode_diff.py); every pose, heading vector and ray column is bit-identical either way: profiles/r06_notes.md.)
DEV float heading_wrap(float th) { return wrap_to_pi(th + 0.5f * PGD_PI) - 0.5f * PGD_PI; }
DEV float not_zero(float x, float eps) { return fabsf(x) > eps ? x : (x > 0.0f ? eps : -eps); }

// counter RNG (IDM timer reseed idm_policy.py:239, scenario resampling base_env.py:451-458)
DEV uint32_t pcg_hash(uint32_t x) {
  uint32_t state = x * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
DEV uint32_t pgd_rng(uint32_t seed, uint32_t a, uint32_t b, uint32_t c) {
  return pcg_hash(seed ^ pcg_hash(a ^ pcg_hash(b ^ pcg_hash(c + 0x9e3779b9u))));
}

// ---------------------------------------------------------------------------------------------------------------------
// lanes: component/lane/straight_lane.py:53-67, circular_lane.py:41-67
// ---------------------------------------------------------------------------------------------------------------------
DEV void lane_local(const pgd_lane& l, float px, float py, float& lon, float& lat) {
  // no fp contraction inside: the own-lane coordinate is carried in the vehicle record AND re-derived by k_derive after
  // pgd_set_state; both must give the same bits, whatever the compiler fuses or re-associates around the inlined copy
#pragma clang fp contract(off)
#pragma clang fp reassociate(off)
  // everything either branch reads goes out before the branch: read field by field a record in memory costs one wait for `dir` and
  // a second one for the fields of the branch taken
  const float4 h = *reinterpret_cast<const float4*>(&l.ax);  // (ax, ay, bx, by)
  const float dir = l.dir;
  float dx = px - h.x, dy = py - h.y;
  if (dir == 0.0f) {
    lon = dx * h.z + dy * h.w;
    lat = dy * h.z - dx * h.w;
  } else {
    float R = h.z, p0 = h.w;
    float phi = p0 + wrap_to_pi(atan2f(dy, dx) - p0);
    lon = dir * (phi - p0) * R;
    lat = dir * (R - sqrtf(dx * dx + dy * dy));
  }
}
DEV void lane_position(const pgd_lane& l, float lon, float lat, float& x, float& y) {
  if (l.dir == 0.0f) {
    x = l.ax + lon * l.bx - lat * l.by;
    y = l.ay + lon * l.by + lat * l.bx;
  } else {
    float phi = l.dir * lon / l.bx + l.by;
    float r = l.bx - lat * l.dir;
    float s, c;
    sincosf(phi, &s, &c);
    x = l.ax + r * c;
    y = l.ay + r * s;
  }
}
DEV float lane_heading_at(const pgd_lane& l, float lon) {
  if (l.dir == 0.0f) return l.c;
  return l.dir * lon / l.bx + l.by + 0.5f * PGD_PI * l.dir;
}
DEV bool lane_is_prev_of(const pgd_lane& a, int b) {  // abs_lane.py:114-119 via the successor table
  bool r = false;
#pragma unroll
  for (int k = 0; k < PGD_MAX_SUCC; ++k) r = r || (k < a.n_succ && a.succ[k] == b);
  return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// oriented boxes
// ---------------------------------------------------------------------------------------------------------------------
struct Obb {
  float cx, cy, ux, uy, hl, hw;
};
DEV Obb obb_of(const pgd_box& b) { return Obb{b.cx, b.cy, b.ux, b.uy, b.hl, b.hw}; }
DEV bool point_in_obb(const Obb& o, float px, float py) {
  float dx = px - o.cx, dy = py - o.cy;
  return fabsf(dx * o.ux + dy * o.uy) <= o.hl && fabsf(dy * o.ux - dx * o.uy) <= o.hw;
}
// ALL: every axis, no early exit -- for waves whose 64 lanes test 64 different pairs within reach of each other (the contact
// lists): they leave at every exit anyway, at an exec-mask switch each.  The line / sidewalk tests keep the exits: most boxes of a
// cell are far from the car and the whole wave leaves at the first axis.
template <bool ALL = false>
DEV bool obb_overlap(const Obb& A, const Obb& B) {  // separating axis test, closed rectangles
  float dx = B.cx - A.cx, dy = B.cy - A.cy;
  float ac = fabsf(A.ux * B.ux + A.uy * B.uy), as = fabsf(A.ux * B.uy - A.uy * B.ux);
  if (ALL) {
    const bool s0 = fabsf(dx * A.ux + dy * A.uy) > A.hl + B.hl * ac + B.hw * as;
    const bool s1 = fabsf(dy * A.ux - dx * A.uy) > A.hw + B.hl * as + B.hw * ac;
    const bool s2 = fabsf(dx * B.ux + dy * B.uy) > B.hl + A.hl * ac + A.hw * as;
    const bool s3 = fabsf(dy * B.ux - dx * B.uy) > B.hw + A.hl * as + A.hw * ac;
    return !(s0 | s1 | s2 | s3);
  }
  if (fabsf(dx * A.ux + dy * A.uy) > A.hl + B.hl * ac + B.hw * as) return false;
  if (fabsf(dy * A.ux - dx * A.uy) > A.hw + B.hl * as + B.hw * ac) return false;
  if (fabsf(dx * B.ux + dy * B.uy) > B.hl + A.hl * ac + A.hw * as) return false;
  if (fabsf(dy * B.ux - dx * B.uy) > B.hw + A.hl * as + A.hw * ac) return false;
  return true;
}
DEV float point_obb_dist(const Obb& o, float px, float py) {  // lidar broad phase (lidar.py:109-124)
  float dx = px - o.cx, dy = py - o.cy;
  float a = fmaxf(fabsf(dx * o.ux + dy * o.uy) - o.hl, 0.0f), c = fmaxf(fabsf(dy * o.ux - dx * o.uy) - o.hw, 0.0f);
  return sqrtf(a * a + c * c);
}
DEV float ray_obb(const Obb& o, float px, float py, float dx, float dy);
// Bodies of the world are chassis / barrier boxes or, for traffic cones and warning tripods (PGD_OBJ_CYLINDER), circles:
// a circle is carried as an Obb with hw < 0 and radius hl.
// OBJ is a compile-time switch: engines whose scenarios hold no traffic objects run kernels without the circle paths.
template <bool OBJ>
DEV bool shape_is_circle(const Obb& o) { return OBJ && o.hw < 0.0f; }
template <bool OBJ, bool ALL = false>
DEV bool shape_overlap(const Obb& box, const Obb& other) {
  if (!shape_is_circle<OBJ>(other)) return obb_overlap<ALL>(box, other);
  return point_obb_dist(box, other.cx, other.cy) <= other.hl;
}
template <bool OBJ>
DEV float shape_point_dist(const Obb& o, float px, float py) {
  if (!shape_is_circle<OBJ>(o)) return point_obb_dist(o, px, py);
  float dx = px - o.cx, dy = py - o.cy;
  return fmaxf(sqrtf(dx * dx + dy * dy) - o.hl, 0.0f);
}
template <bool OBJ>
DEV float shape_ray(const Obb& o, float px, float py, float dx, float dy) {
  if (!shape_is_circle<OBJ>(o)) return ray_obb(o, px, py, dx, dy);
  // |p + t d - c|^2 = r^2, smallest root in (0, 1]; a ray that starts inside does not hit (as for boxes).  Written with
  // the unit direction u: offset of the centre across the ray = cross(f, u) (no cancellation, unlike b^2 - a c), along it
  // = dot(f, u); hit distance = -dot - sqrt(r^2 - cross^2)
  const float fx = px - o.cx, fy = py - o.cy;
  const float len = sqrtf(dx * dx + dy * dy);
  if (fx * fx + fy * fy <= o.hl * o.hl || len <= 0.0f) return 1.0f;
  const float ux = dx / len, uy = dy / len;
  const float along = fx * ux + fy * uy, across = fx * uy - fy * ux;
  const float h2 = o.hl * o.hl - across * across;
  if (h2 < 0.0f || along >= 0.0f) return 1.0f;
  const float t = (-along - sqrtf(h2)) / len;
  return (t > 0.0f && t <= 1.0f) ? t : 1.0f;
}
// nearest hit fraction of the segment p + t d, t in [0,1]; 1 = miss (cutils.pyx:60-142 rayTestClosest)
DEV float ray_obb(const Obb& o, float px, float py, float dx, float dy) {
  float rx = px - o.cx, ry = py - o.cy;
  float ox = rx * o.ux + ry * o.uy, oy = ry * o.ux - rx * o.uy;
  float vx = dx * o.ux + dy * o.uy, vy = dy * o.ux - dx * o.uy;
  // the two slabs without a branch (the same operations on the same operands as the branching form: a ray parallel to a slab --
  // |v| < 1e-12 -- leaves the interval as it is when the origin lies inside the slab and empties it otherwise): a wave that casts
  // 64 different (ray, box) incidences took every path of the branching form anyway, at four exec-mask switches each
  const bool par_x = fabsf(vx) < 1e-12f, par_y = fabsf(vy) < 1e-12f;
  const float ix = 1.0f / (par_x ? 1.0f : vx), iy = 1.0f / (par_y ? 1.0f : vy);
  const float ax = (-o.hl - ox) * ix, bx = (o.hl - ox) * ix, ay = (-o.hw - oy) * iy, by = (o.hw - oy) * iy;
  const float lox = par_x ? (fabsf(ox) > o.hl ? 2.0f : 0.0f) : fminf(ax, bx), hix = par_x ? 1.0f : fmaxf(ax, bx);
  const float loy = par_y ? (fabsf(oy) > o.hw ? 2.0f : 0.0f) : fminf(ay, by), hiy = par_y ? 1.0f : fmaxf(ay, by);
  const float t0 = fmaxf(fmaxf(0.0f, lox), loy), t1 = fminf(fminf(1.0f, hix), hiy);
  return (t0 <= t1 && t0 > 0.0f) ? t0 : 1.0f;  // origin inside the box: no hit (Bullet's convex ray cast semantics)
}

// BaseVehicle.projection (base_vehicle.py:460-475)
DEV void projection(float hx, float hy, float vx, float vy, float& ph, float& ps) {
  float l = norm2(hx, hy);
  ph = (vx * hx + vy * hy) / (l + 1e-6f);
  float sx = -hy / l, sy = hx / l;
  ps = (vx * sx + vy * sy) / (norm2(sx, sy) + 1e-6f);
}

// the scripted lane-keeping policy (pgd_lane_keep_actions / pgd_step_lane_keep; include/pgdrive_hip.h): ONE arithmetic for the
// stand-alone kernel and the copy inside k_step (fused multiply-adds written out, no contraction: the two must give the same bits)
DEV float2 lane_keep_action(const uint32_t seed, const int env_global, const float o0, const float o1, const float o2, const float o3, const float k_lat,
                            const float k_head, const float v_target, const float noise, const uint32_t tick) {
#pragma clang fp contract(off)
#pragma clang fp reassociate(off)
  const uint32_t r = pgd_rng(seed ^ 0x1a7e5eedu, (uint32_t)env_global, 0x900dcafeu, tick);
  const float n1 = fmaf((float)(r & 0xffffu), 2.0f / 65535.0f, -1.0f), n2 = fmaf((float)(r >> 16), 2.0f / 65535.0f, -1.0f);
  const float v_kmh = fmaf(o3, 81.0f, -1.0f);  // state_obs.py:82: (speed + 1) / (max_speed + 1), max_speed 80 km/h
  const float st = fmaf(noise, n1, fmaf(k_head, fmaf(2.0f, o2, -1.0f), (k_lat * 1.8f) * (o0 - o1)));
  const float tb = fmaf(noise, n2, 0.3f * (v_target - v_kmh));
  return make_float2(clipf(st, -1.0f, 1.0f), clipf(tb, -1.0f, 1.0f));
}

DEV float pid_update(float& p, float& i, float kp, float ki, float kd, float err) {  // PID_controller.py:10-17
  i += err;
  float dd = err - p;
  p = err;
  return -kp * p - ki * i - kd * dd;
}
