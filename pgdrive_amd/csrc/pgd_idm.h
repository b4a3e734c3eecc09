// pgd_idm.h -- IDM traffic policy: routing, neighbour search, lane change, PID steering, IDM law.
// Part of the single translation unit pgd_engine.hip (included there, in this order, after pgd_device.h).
#ifndef PGD_IDM_H
#define PGD_IDM_H

// ---------------------------------------------------------------------------------------------------------------------
// IDM: policy/idm_policy.py:82-133 (FrontBackObjects), :190-353 (act, lane change), :244-271 (PID + IDM law)
// ---------------------------------------------------------------------------------------------------------------------
#ifndef PGD_RELMASK_MIN
#define PGD_RELMASK_MIN 3  // bodies in some vehicle's broad phase from which the related-lane pass of find_front_back runs
#endif
struct Fbo {
  int front[3], back[3];
  float fd[3], bd[3];
  bool exist[3];
};

// own_*: the vehicle's coordinates on `lane` and the lane heading one metre ahead, by-products of the search on the own lane
// (target 1), handed to every sub-lane: the steering controller needs exactly these when it steers along `lane`.
// `idx` / `n_road`: index of `lane` in its road and that road's lane count (the lanes of a road are consecutive); only read when
// `with_ref` holds -- the caller then has both from the vehicle's route context (no lane-table read)
DEV void find_front_back(const MapView& mv, const Grp& g, const Snap& S, int base, int V, int self, unsigned long long objs,
                         int lane, int idx, int n_road, float max_dist, bool with_ref, Fbo& r, float& own_lon, float& own_lat, float& own_head) {
  const int l0 = (with_ref && idx > 0) ? lane - 1 : -1;
  const int l2 = (with_ref && idx + 1 < n_road) ? lane + 1 : -1;
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
    // (round 6: this record read ahead by the sub-lane, next to the snapshot's own lane read -- one memory round trip less on paper --
    // made every row SLOWER, metric 17.02 -> 17.22 us, respawn 24.07 -> 24.30: like every read put in front of the first chain before it)
    const pgd_lane& li = mv.lanes[tl];
    float cur, lat;
    lane_local(li, px, py, cur, lat);
    if (t == 1) { own_lon = cur; own_lat = lat; own_head = lane_heading_at(li, cur + 1.0f); }
    const float left_long = li.length - cur;
    const int4 lsucc = *reinterpret_cast<const int4*>(li.succ);
    // one pass, five running minima (FrontBackObjects.get_find_front_back_objs, idm_policy.py:107-131):
    //   same lane front/back; successor-lane front; predecessor-lane back (all / excluding successor-lane objects)
    float same_f = max_dist, same_b = max_dist, succ_f = max_dist, pred_b = max_dist, pred_bx = max_dist;
    int o_same_f = -1, o_same_b = -1, o_succ_f = -1, o_pred_b = -1, o_pred_bx = -1;
    bool found_f = false, found_b = false;
    // Only a body on the target lane, on one of its successors or on one of its predecessors can count; the device copy of the
    // lane carries that set as a 32-bit mask over (lane id & 31) (pgd_upload_maps).  One uniform pass over the slots -- every
    // lane of the wave reads the same S.lane[o]: broadcasts, no dependent chain -- leaves the few bodies the search proper
    // has to look at: the loop below costs ~60 instructions and four LDS reads per body, and the wave pays for the longest
    // list among its lanes (dense traffic: the broad phase holds 8 - 16 bodies, 1 - 3 of them on a related lane).
    unsigned long long rel_objs = objs;
#ifndef PGD_NO_RELMASK
    // (wave-uniform switch: with short lists everywhere -- the metric's workload -- the pass would cost more than it saves)
    if (__ballot(__popcll(objs) >= PGD_RELMASK_MIN) != 0ull) {
      const unsigned rel = __float_as_uint(li.ey);
      unsigned lo_m = 0u, hi_m = 0u;
      const int v_lo = V < 32 ? V : 32;
      for (int o0 = 0; o0 < v_lo; o0 += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)  // (slots past the last one repeat it: their bits are not in `objs`)
          lo_m |= ((rel >> (S.lane[base + min(o0 + j, V - 1)] & 31)) & 1u) << ((o0 + j) & 31);
      }
      for (int o = 32; o < V; ++o) hi_m |= ((rel >> (S.lane[base + o] & 31)) & 1u) << (o - 32);
      rel_objs &= ((unsigned long long)hi_m << 32) | lo_m;
    }
#endif
    // only the vehicles inside the broad phase, in slot order (ties keep the first one like the reference's loop)
    for (unsigned long long m = rel_objs; m != 0ull; m &= m - 1ull) {
      const int o = __builtin_ctzll(m);
      const int ol = S.lane[base + o];
      const float olon = S.lon[base + o], ollen = S.llen[base + o];
      const int4 osucc = S.succ[base + o];
      // the five running minima by selects: the search lanes of a wave (three target lanes per vehicle) meet the same body on
      // different kinds of lane -- its own, a successor, a predecessor, none -- and took every branch of the nested form one after
      // the other; the same comparisons on the same values, predicated
      const bool same = ol == tl;
      const bool is_succ = !same && succ_has(lsucc, ol);
      const bool is_pred = !same && succ_has(osucc, tl);
      const float lg_s = olon - cur, lg_f = olon + left_long, lg_p = ollen - olon + cur;
      const bool u0 = same && same_f > lg_s && lg_s > 0.0f;
      const bool u1 = same && lg_s < 0.0f && fabsf(lg_s) < same_b;
      const bool u2 = is_succ && succ_f > lg_f && lg_f > 0.0f;
      const bool u3 = is_pred && pred_b > lg_p;
      const bool u4 = is_pred && !is_succ && pred_bx > lg_p;
      same_f = u0 ? lg_s : same_f; o_same_f = u0 ? o : o_same_f; found_f = found_f || u0;
      same_b = u1 ? fabsf(lg_s) : same_b; o_same_b = u1 ? o : o_same_b; found_b = found_b || u1;
      succ_f = u2 ? lg_f : succ_f; o_succ_f = u2 ? o : o_succ_f;
      pred_b = u3 ? lg_p : pred_b; o_pred_b = u3 ? o : o_pred_b;
      pred_bx = u4 ? lg_p : pred_bx; o_pred_bx = u4 ? o : o_pred_bx;
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
  {
    const int src = g.lead + (1 % g.SUB);
    own_lon = __shfl(own_lon, src); own_lat = __shfl(own_lat, src); own_head = __shfl(own_head, src);
  }
}

template <bool OBJ>
DEV void idm_act(const PgdDev& d, const MapView& mv, const Grp& g, const pgd_spawn& sp, const Snap& S, int base, int V,
                 int s, int e, uint32_t step_count, Veh& r, float& out_steer, float& out_acc) {
  const float NORMAL = 30.0f, CREEP = 5.0f, SAFE = 15.0f, MAXD = 30.0f;
  const int vlane = r.lane;
  int rt = r.rlane;
  // "the lane lies on the current road": the lanes of a road are consecutive ids, and the record's route context carries the first
  // lane and the lane count of the current road -- a range test instead of reading the lanes' road ids from the lane table (two
  // dependent global reads at the head of every IDM decision; with no current road the count is 0 and the test fails, as the
  // comparison of road ids did)
  struct { int first_lane, n_lanes; } const CR{r.cur_first, r.cur_n};
  const bool vl_on_cur = vlane >= CR.first_lane && vlane < CR.first_lane + CR.n_lanes;
  const bool rt_on_cur = rt >= CR.first_lane && rt < CR.first_lane + CR.n_lanes;
  bool success;
  // move_to_next_road (idm_policy.py:222-242)
  if (rt < 0) {
    rt = vlane;
    success = vl_on_cur;
  } else if (!rt_on_cur) {
    // the lowest lane of the current road that follows the routing lane (8 packed successor ids, unused = -1)
    const int4 rs = *reinterpret_cast<const int4*>(mv.lanes[rt].succ);
    const int first = CR.first_lane, nl = CR.n_lanes;
    int best = 0x7fff;
    const int w4[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int lo = (int)(short)(w4[q] & 0xffff) - first, hi = (w4[q] >> 16) - first;
      if (lo >= 0 && lo < nl) best = min(best, lo);
      if (hi >= 0 && hi < nl) best = min(best, hi);
    }
    success = best != 0x7fff;
    if (success) rt = first + best;
  } else if (vl_on_cur && rt != vlane) {
    rt = vlane;
    r.timer = (pgd_rng(d.cfg.seed, (uint32_t)(d.cfg.env_base + e), (uint32_t)s, step_count) % 25u);
    success = true;
  } else success = true;
  // is the (new) routing lane on the current road?  first case: the vehicle lane's road; second: only a found lane of
  // CR; third and fourth: established by the branch conditions
  const bool in_cur = r.rlane < 0 ? vl_on_cur : (!rt_on_cur ? success : true);
  r.rlane = rt;

  // Lidar.get_surrounding_objects (lidar.py:109-124)
  float px = S.x[base + s], py = S.y[base + s];
  unsigned long long objs = 0ull;
  // the sub-lanes of the vehicle split the slots (o = sub, sub + SUB, ...) and merge their bit sets: ceil(V / SUB) distance
  // tests per lane whatever the number of bodies in the world
#ifdef PGD_EXITAT
  if (!PGD_DBG_SKIP(0))
#endif
  // a trip count that does not depend on the sub-lane (a constant in the kernels specialised for a default configuration): the
  // rounds unroll, their LDS reads go out together and the distance tests of different slots overlap
  {
    const int rounds = (V + g.SUB - 1) / g.SUB;
#pragma unroll 6
    for (int j = 0; j < rounds; ++j) {
      const int o = g.sub + j * g.SUB, oc = min(o, V - 1);
      const Obb ob = snap_obb(S, base + oc);
      const bool in = o < V && o != s && S.present[base + oc] && shape_point_dist<OBJ>(ob, px, py) <= 50.0f;
      objs |= in ? (1ull << oc) : 0ull;
    }
  }
  {
    unsigned lo = (unsigned)objs, hi = (unsigned)(objs >> 32);
    lo = group_or(lo, g);
    hi = V > 32 ? group_or(hi, g) : 0u;
    objs = ((unsigned long long)hi << 32) | lo;
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
  float own_lon = 0.0f, own_lat = 0.0f, own_head = 0.0f;
#ifdef PGD_EXITAT
  if (PGD_DBG_SKIP(1)) { for (int i = 0; i < 3; ++i) { fb.front[i] = fb.back[i] = -1; fb.fd[i] = fb.bd[i] = MAXD; fb.exist[i] = true; } } else
#endif
  // (with the reference lanes -- success -- the search only runs when the routing lane is on the current road: its index there)
  if (search) find_front_back(mv, g, S, base, V, s, objs, rt, rt - CR.first_lane, CR.n_lanes, MAXD, success, fb, own_lon, own_lat, own_head);
  PHASE_MARK(10);  // idm: front/back search
#ifdef PGD_EXITAT
  if (PGD_DBG_SKIP(2)) { } else
#endif
  if (success && in_cur) {
    int idx = rt - CR.first_lane;
    int n_cur = CR.n_lanes;
    int avail_lo = 0, avail_hi = n_cur - 1;
    bool decided = false;
    if (r.ck0 != r.ck1) {
      struct { int first_lane, n_lanes; } const NR{r.next_first, r.next_n};
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
        r.timer = min((int)r.timer + 1, 0xffff);
        front_obj = fb.front[1]; front_dist = fb.fd[1]; steer_lane = rt;
      }
    }
  } else if (!success) {
    front_obj = fb.front[1]; front_dist = fb.fd[1]; steer_lane = rt;
  }
  PHASE_MARK(11);  // idm: lane-change logic

  // steering_control (idm_policy.py:244-252)
#ifdef PGD_EXITAT
  if (PGD_DBG_SKIP(3)) { out_steer = 0.0f; out_acc = 0.0f; return; }
#endif
  float lon, lat, lane_heading;
  if (search && steer_lane == rt) {  // the search on the own lane has evaluated exactly this (same routine, same inputs)
    lon = own_lon; lat = own_lat; lane_heading = own_head;
  } else {
    const pgd_lane& SL = mv.lanes[steer_lane];
    lane_local(SL, px, py, lon, lat);
    lane_heading = lane_heading_at(SL, lon + 1.0f);
  }
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

#endif
