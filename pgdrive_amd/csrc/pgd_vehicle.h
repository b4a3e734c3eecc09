// pgd_vehicle.h -- per-lane vehicle registers, the LDS snapshot of an env's vehicles, sub-lane groups.
// Part of the single translation unit pgd_engine.hip (included there, in this order, after pgd_device.h).
#ifndef PGD_VEHICLE_H
#define PGD_VEHICLE_H

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

DEV void load_rec(const VehRec* rec, Veh& r) {
  VehRec t;
  const uint4* src = reinterpret_cast<const uint4*>(rec);
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
DEV void load_veh(const PgdDev& d, int e, int s, Veh& r) { load_rec(d.rec + (size_t)e * d.V + s, r); }
DEV void store_rec(VehRec* rec, const Veh& r) {
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
  uint4* dst = reinterpret_cast<uint4*>(rec);
  const uint4* src = reinterpret_cast<const uint4*>(&t);
#pragma unroll
  for (int k = 0; k < 8; ++k) dst[k] = src[k];
}
DEV void store_veh(const PgdDev& d, int e, int s, const Veh& r) { store_rec(d.rec + (size_t)e * d.V + s, r); }

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

#endif
