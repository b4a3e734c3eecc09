// pgd_vehicle.h -- per-lane vehicle registers, the LDS snapshot of an env's vehicles, sub-lane groups.
// Part of the single translation unit pgd_engine.hip (included there, in this order, after pgd_device.h).
#ifndef PGD_VEHICLE_H
#define PGD_VEHICLE_H

// ---------------------------------------------------------------------------------------------------------------------
// per-lane vehicle registers: struct Veh (pgd_device.h) is the record itself
// ---------------------------------------------------------------------------------------------------------------------
// records in memory: blocks of n records as eight planes of n pieces (RecPiece, pgd_device.h)
DEV const RecPiece* rec_block(const RecPiece* base, size_t block, int n) { return base + block * (size_t)(8 * n); }
DEV RecPiece* rec_block(RecPiece* base, size_t block, int n) { return base + block * (size_t)(8 * n); }
DEV void load_rec(const RecPiece* blk, int n, int s, Veh& r) {
  uint4* dst = reinterpret_cast<uint4*>(&r);
#pragma unroll
  for (int k = 0; k < 8; ++k) dst[k] = blk[k * n + s].q;
}
// the first 64 bytes: pose, speed, heading vector, lane | spawn, status / flags, route words (what an observer needs of a body)
DEV void load_rec_head(const RecPiece* blk, int n, int s, Veh& r) {
  uint4* dst = reinterpret_cast<uint4*>(&r);
#pragma unroll
  for (int k = 0; k < 4; ++k) dst[k] = blk[k * n + s].q;
}
DEV void store_rec(RecPiece* blk, int n, int s, const Veh& r) {
  const uint4* src = reinterpret_cast<const uint4*>(&r);
#pragma unroll
  for (int k = 0; k < 8; ++k) blk[k * n + s].q = src[k];
}
// 32-bit word w (0 .. 31) of record s: struct Veh's fields by their word index
DEV uint32_t rec_word(const RecPiece* blk, int n, int s, int w) { return reinterpret_cast<const uint32_t*>(blk + (w >> 2) * n + s)[w & 3]; }
DEV float rec_float(const RecPiece* blk, int n, int s, int w) { return __uint_as_float(rec_word(blk, n, s, w)); }
enum { RW_X = 0, RW_Y = 1, RW_TH = 2, RW_LANE_SPAWN = 8 };
DEV int rec_spawn(const RecPiece* blk, int n, int s) { return (int)(rec_word(blk, n, s, RW_LANE_SPAWN) >> 16); }  // Veh::spawn
DEV void load_veh(const PgdDev& d, int e, int s, Veh& r) { load_rec(rec_block(d.rec, (size_t)e, d.V), d.V, s, r); }
DEV void store_veh(const PgdDev& d, int e, int s, const Veh& r) { store_rec(rec_block(d.rec, (size_t)e, d.V), d.V, s, r); }

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

// Poses of every vehicle after the sub-steps of the physics BEFORE the last one (the last is the snapshot itself): contacts are
// raised inside each doPhysics call (engine_core.py:276-278, collision_callback.py:7-36).  Lives in the same LDS bytes as
// ObsLds (union ObsScratch, pgd_observe.h): the contact test is over before the observation scratch is first written.
#define PGD_MAX_SUB 5
#ifndef PGD_SUBV
#define PGD_SUBV 52  // slots of a wave with sub-step poses (the largest shipped env: 40 agents + 8 toll booths; three packed
                     // envs of 17 slots); waves with more slots test contacts at the end pose only
#endif
struct SubPose {
  float4 p[PGD_MAX_SUB - 1][PGD_SUBV];  // (x, y, unit vector of the MOTION direction) of the slot after sub-step k + 1
  float2 beta[PGD_SUBV];                // (cos, sin) of the slip angle: heading = motion direction rotated back by it
  float trav[PGD_SUBV];                 // path length of the slot in this step (0 = did not move: p[] is not written)
};

// ---------------------------------------------------------------------------------------------------------------------
// sub-lane cooperation: a vehicle is carried by SUB consecutive lanes that hold identical copies of its registers; the
// heavy box / neighbour loops are split across them and recombined with wave shuffles (all lanes of a group are always
// convergent because they execute on identical data).
// ---------------------------------------------------------------------------------------------------------------------
struct Grp {
  int sub, SUB, lead;
};
// up to 4 sub-lanes (V >= 16): four INDEPENDENT shuffles (clamped index) that pipeline, instead of a dependent loop
DEV unsigned group_min(unsigned v, const Grp& g) {
  if (g.SUB <= 4) {
    const int m = g.SUB - 1;
    const unsigned a = (unsigned)__shfl((int)v, g.lead), b = (unsigned)__shfl((int)v, g.lead + min(1, m)),
                   c = (unsigned)__shfl((int)v, g.lead + min(2, m)), e = (unsigned)__shfl((int)v, g.lead + m);
    return min(min(a, b), min(c, e));
  }
  unsigned r = v;
  for (int j = 0; j < g.SUB; ++j) r = min(r, (unsigned)__shfl((int)v, g.lead + j));
  return r;
}
DEV unsigned group_or(unsigned v, const Grp& g) {
  if (g.SUB <= 4) {
    const int m = g.SUB - 1;
    const unsigned a = (unsigned)__shfl((int)v, g.lead), b = (unsigned)__shfl((int)v, g.lead + min(1, m)),
                   c = (unsigned)__shfl((int)v, g.lead + min(2, m)), e = (unsigned)__shfl((int)v, g.lead + m);
    return a | b | c | e;
  }
  unsigned r = v;
  for (int j = 0; j < g.SUB; ++j) r |= (unsigned)__shfl((int)v, g.lead + j);
  return r;
}

#endif
