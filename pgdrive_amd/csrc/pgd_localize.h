// pgd_localize.h -- lane localisation, checkpoints, line / sidewalk contacts, route context, after_step.
// Part of the single translation unit pgd_engine.hip (included there, in this order, after pgd_device.h).
#ifndef PGD_LOCALIZE_H
#define PGD_LOCALIZE_H

// ---------------------------------------------------------------------------------------------------------------------
// localisation: utils/scene_utils.py:138-185 + navigation.py:328-344.  "First hit" = smallest box id (Bullet insertion
// order); the cell-major box copies keep that order, so the smallest list position per class is the answer.
// key = (position in cell << 16) | lane id
// ---------------------------------------------------------------------------------------------------------------------
// Device-side cell index (built by pgd_upload_maps): inside a cell the lane-surface boxes come first (original relative
// order), the line / sidewalk boxes follow.  cstart[c] = first item | (number of lane boxes << 24); the cell ends where the
// next one starts.  Localisation scans only the lane part, the contact / ray tests only the rest.
DEV int cell_first(int c) { return c & 0xffffff; }
DEV int cell_mid(int c) { return (c & 0xffffff) + (int)((unsigned)c >> 24); }

// nearest hit fraction of the segment p + t d (t in [0,1]) against the map's boxes whose kind is in `kinds`: Amanatides-Woo
// walk over the uniform grid; a cell is skipped once its entry parameter is beyond the best hit so far
template <class MV>
DEV float ray_grid(const MV& mv, float px, float py, float dx, float dy, unsigned kinds) {
  const float inv = 1.0f / mv.cell();
  int ix = (int)floorf((px - mv.ox()) * inv), iy = (int)floorf((py - mv.oy()) * inv);
  const int sx = dx > 0.0f ? 1 : -1, sy = dy > 0.0f ? 1 : -1;
  const float big = 3.0e38f;
  const float tdx = dx != 0.0f ? fabsf(mv.cell() / dx) : big, tdy = dy != 0.0f ? fabsf(mv.cell() / dy) : big;
  float tmx = dx != 0.0f ? ((mv.ox() + (ix + (dx > 0.0f ? 1 : 0)) * mv.cell()) - px) / dx : big;
  float tmy = dy != 0.0f ? ((mv.oy() + (iy + (dy > 0.0f ? 1 : 0)) * mv.cell()) - py) / dy : big;
  float best = 1.0f, t_enter = 0.0f;
  for (int it = 0; it < 64; ++it) {
    if (t_enter > best + 0.02f) break;  // boxes are registered with a 5 cm margin: keep a little slack
    if (ix >= 0 && iy >= 0 && ix < mv.gx() && iy < mv.gy()) {
      const int cell = iy * mv.gx() + ix;
      const int k1 = cell_first(mv.cstart[cell + 1]);
      for (int k = cell_mid(mv.cstart[cell]); k < k1; ++k) {
        const pgd_box b = mv.cbox[k];
        if (!((1u << b.kind) & kinds)) continue;
        best = fminf(best, ray_obb(obb_of(b), px, py, dx, dy));
      }
    } else if ((sx > 0 ? ix >= mv.gx() : ix < 0) || (sy > 0 ? iy >= mv.gy() : iy < 0)) {
      break;  // left the grid for good
    }
    if (tmx < tmy) { t_enter = tmx; tmx += tdx; ix += sx; }
    else { t_enter = tmy; tmy += tdy; iy += sy; }
    if (t_enter > 1.0f) break;
  }
  return best;
}


// cell_start entry of the grid cell under (px, py); 0 (an empty range) outside the grid
template <class MV>
DEV int cell_entry(const MV& mv, float px, float py) {
  int cx = (int)floorf((px - mv.ox()) / mv.cell()), cy = (int)floorf((py - mv.oy()) / mv.cell());
  return (cx >= 0 && cy >= 0 && cx < mv.gx() && cy < mv.gy()) ? mv.cstart[cy * mv.gx() + cx] : 0;
}

// `c` = cell_entry(mv, px, py), read by the caller ahead of time
DEV int get_current_lane(const MapView& mv, const Grp& g, int c, float px, float py, float hx, float hy, int road_cur,
                         int road_next) {
  const int k0 = cell_first(c), k1 = cell_mid(c);
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

// Route context of the vehicle (Navigation.current_ref_lanes / next_ref_lanes, navigation.py:155-183): roads of the current
// and the next checkpoint pair, their first lanes and lane counts, block id of the current road.  Kept in the vehicle record
// and rebuilt only here -- when a checkpoint is passed, at (re)spawn and by k_derive -- so that localisation, the IDM routing,
// side distances, reward and observation read it from registers instead of walking spawn record -> road table every step.
DEV void route_refresh(const MapView& mv, const pgd_spawn& sp, Veh& r) {
  const int rc = sp.ckpt_road[r.ck0], rn = sp.ckpt_road[r.ck1];
  r.road_cur = (uint32_t)rc; r.road_next = (uint32_t)rn;  // -1 (no road) becomes 0xfff, which no map uses (<= 4095 roads)
  r.blk = 0; r.cur_first = 0; r.cur_n = 0; r.next_first = 0; r.next_n = 0;
  if (rc >= 0) {
    const pgd_road& CR = mv.roads()[rc];
    r.blk = CR.block_id; r.cur_first = CR.first_lane; r.cur_n = CR.n_lanes;
  }
  if (rn >= 0) {
    const pgd_road& NR = mv.roads()[rn];
    r.next_first = NR.first_lane; r.next_n = NR.n_lanes;
  }
}

// Navigation._update_target_checkpoints (navigation.py:262-282)
// `start_node`: from-node of the road of the vehicle's lane (the device lane copy carries it: pgd_lane::pad, pgd_upload_maps).
// The device copy of a spawn record holds PGD_CKPT_END in the route from its LAST node on (pgd_upload_scenarios): the search below --
// checkpoints[ck1:].index(start_node) with index < len - 1 (navigation.py:270-277) -- then needs neither the route length nor the
// lane -> road table chain in front of its reads: one memory round trip on every step of the first five metres of a lane where
// rounds 1 - 5 made three dependent ones (lane record, road record, route; PGD_CKPT_CHAIN keeps that form for A/B).
#define PGD_CKPT_END (-32768)
DEV void update_checkpoints(const MapView& mv, const Grp& g, const pgd_spawn& sp, Veh& r, float lon, const int start_node) {
  if (r.ck0 == r.ck1) return;
  if (!(lon < 5.0f)) return;
  // The first match decides, and a match on the last node alone changes nothing.  The sub-lanes of the vehicle split the tail,
  // four independent reads each per round (a whole route in one round trip), and take the lowest hit.
  unsigned hit = 0xffffffffu;
  const int step = g.SUB;
#ifdef PGD_CKPT_CHAIN
  const int n = sp.n_ckpt;
  const int start_node_ = mv.roads()[mv.lanes[r.lane].road].from;
  for (int k = r.ck1 + g.sub; k < n - 1; k += 4 * step) {
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = sp.ckpt[min(k + j * step, n - 2)];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k + j * step < n - 1 && v[j] == start_node_) hit = min(hit, (unsigned)(k + j * step));
  }
#else
  for (int k = r.ck1 + g.sub; k < PGD_MAX_CKPT; k += 4 * step) {
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = sp.ckpt[min(k + j * step, PGD_MAX_CKPT - 1)];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k + j * step < PGD_MAX_CKPT && v[j] == start_node) hit = min(hit, (unsigned)(k + j * step));
    if (v[3] == PGD_CKPT_END) break;  // (the end marks are contiguous: this sub-lane's later entries are marks as well)
  }
#endif
  hit = group_min(hit, g);
  if (hit == 0xffffffffu) return;
  const int idx = (int)hit;
  const int n_nodes = sp.n_ckpt;  // (a checkpoint was passed: once per road)
  r.ck0 = idx;
  r.ck1 = (idx + 1 == n_nodes - 1) ? idx : idx + 1;
  route_refresh(mv, sp, r);
}

// Navigation.update_localization (navigation.py:155-183)
// `PL` = the record of r.lane, read by the caller (ahead of time, together with its other lane reads)
template <class MV>
DEV void update_localization(const MV& mv, const Grp& g, const pgd_spawn& sp, Veh& r, float& lon_out, float& lat_out,
                             const pgd_lane& PL) {
  const float s = r.hy, c = r.hx;
  const int road_cur = (int)r.road_cur;  // the record's route context: no spawn-record read
  const int road_next = (r.ck0 == r.ck1) ? -1 : (int)r.road_next;
  PHASE_MARK(16);  // after_step: route roads
  // Staying on a lane of the current road needs no grid walk.  The surface box of a lane is (length + 0.1) x
  // (width + 1.2) (base_block.py:396-456), the boxes of a road's lanes overlap by 1.2 m and the first hit in creation order
  // -- lane index order inside the road -- wins, boxes of the current road before all others.  So the previous lane is the
  // answer when the vehicle is inside its box, outside the box of the left neighbour (the only earlier box of the road that
  // can reach it) and heads along it.  Margins of 5 cm keep the decision away from the fp32 rounding of either form.
  float lon = 0.0f, lat = 0.0f;
  bool stay = false;
  const int cell = cell_entry(mv, r.x, r.y);  // in flight together with the lane records of the caller
  {
    if (PL.road == road_cur) {
      lane_local(PL, r.x, r.y, lon, lat);
      const float hw = 0.5f * PL.width;
      if (PL.dir == 0.0f) {
        stay = lon >= 0.0f && lon <= PL.length && lat <= hw + 0.55f && lat >= (PL.index > 0 ? 0.65f - hw : -hw - 0.55f) &&
               PL.bx * c + PL.by * s > 0.0f;
      } else {
        // Arc lanes are covered by n = int(length / 4) boxes of 1.3 x the segment length, each centred ON the arc and
        // turned along the second half of its segment (base_block.py:413-423): within its segment a box axis leaves the arc
        // by less than seg^2 / (4 R).  With that bound (taken on the innermost radius in play, 10 % and 5 cm of slack) the
        // same two statements hold: inside an own box, outside every box of the left neighbour -- whose segments may be
        // longer.  A lane shorter than one segment has no box at all; tight arcs simply take the grid walk.
        const int n = (int)(PL.length * 0.25f);
        const float r_in = PL.bx - PL.width - 1.0f;  // PL.bx: radius
        if (n >= 1 && r_in > 4.0f) {
          const float seg = PL.length / (float)n;
          const float dev = seg * seg / (4.0f * r_in) * 1.1f + 0.05f;
          float lo = -hw - 0.6f + dev;
          if (PL.index > 0) {
            const float nl = mv.lanes[r.lane - 1].length;
            const int nn = (int)(nl * 0.25f);
            if (nn >= 1) {
              const float nseg = nl / (float)nn;
              lo = fmaxf(lo, 0.6f - hw + nseg * nseg / (4.0f * r_in) * 1.1f + 0.05f);
            }
          }
          const float tx = -PL.dir * (r.y - PL.ay), ty = PL.dir * (r.x - PL.ax);  // lane direction at the point (x |radius|)
          stay = lon >= 0.0f && lon <= PL.length && lat <= hw + 0.6f - dev && lat >= lo && tx * c + ty * s > 0.0f;
        }
      }
    }
  }
  int lane = r.lane;
  if (!stay) lane = get_current_lane(mv, g, cell, r.x, r.y, c, s, road_cur, road_next);
  PHASE_MARK(17);  // after_step: get_current_lane
  bool on_lane = lane >= 0;
  if (!on_lane) lane = r.lane;
  r.lane = lane;
  int from_node = PL.pad;  // (the lane of the previous step, still the vehicle's when it stays)
  if (!stay) {
    const pgd_lane& NL = mv.lanes[lane];
    from_node = NL.pad;
    lane_local(NL, r.x, r.y, lon, lat);
  }
  lon_out = lon; lat_out = lat;
  r.lon = lon;
  update_checkpoints(mv, g, sp, r, lon, from_node);
  r.vflags = on_lane ? (r.vflags & ~PGD_F_OFF_LANE) : (r.vflags | PGD_F_OFF_LANE);
  PHASE_MARK(18);  // after_step: lane_local + checkpoints
}

// BaseVehicle._state_check (base_vehicle.py:615-644): the car's box against the line / sidewalk boxes of the grid cells under
// it.  The (<= 2 x 2) cells are flattened into one index range (their four start offsets are read at once: one dependent
// level for the whole neighbourhood) that the sub-lanes of the vehicle stride through, two boxes in flight per lane.
// state_check_part: the share of lane `sub` of `SUB` (any SUB >= 1, the lanes need not be neighbours); the caller ORs the shares
template <class MV>
DEV unsigned state_check_part(const MV& mv, const int g_sub, const int g_SUB, const Obb& car) {
  float ex = fabsf(car.ux) * car.hl + fabsf(car.uy) * car.hw, ey = fabsf(car.uy) * car.hl + fabsf(car.ux) * car.hw;
  int cx0 = max((int)floorf((car.cx - ex - mv.ox()) / mv.cell()), 0), cx1 = min((int)floorf((car.cx + ex - mv.ox()) / mv.cell()), mv.gx() - 1);
  int cy0 = max((int)floorf((car.cy - ey - mv.oy()) / mv.cell()), 0), cy1 = min((int)floorf((car.cy + ey - mv.oy()) / mv.cell()), mv.gy() - 1);
  unsigned fl = 0;
  for (int cyb = cy0; cyb <= cy1; cyb += 2)
    for (int cxb = cx0; cxb <= cx1; cxb += 2) {  // blocks of up to 2x2 cells (a car spans at most 2 cells per axis)
      int k0[4], pre[5];
      pre[0] = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int cx = cxb + (q & 1), cy = cyb + (q >> 1);
        bool in = cx <= cx1 && cy <= cy1;
        int cell = in ? cy * mv.gx() + cx : 0;
        int a = cell_mid(mv.cstart[cell]), b = cell_first(mv.cstart[cell + 1]);
        k0[q] = a;
        pre[q + 1] = pre[q] + (in ? b - a : 0);
      }
      const int n = pre[4];
      for (int f = g_sub; f < n; f += 2 * g_SUB) {
        pgd_box b[2];
        bool have[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int fj = f + j * g_SUB;
          have[j] = fj < n;
          const int ff = have[j] ? fj : f;
          const int q = (ff >= pre[1]) + (ff >= pre[2]) + (ff >= pre[3]);
          const int kk = (q == 0 ? k0[0] : q == 1 ? k0[1] : q == 2 ? k0[2] : k0[3]) + ff - (q == 0 ? pre[0] : q == 1 ? pre[1] : q == 2 ? pre[2] : pre[3]);
          b[j] = mv.cbox[kk];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (!have[j]) continue;
          const unsigned bit = b[j].kind == PGD_BOX_WHITE ? PGD_F_ON_WHITE
                               : b[j].kind == PGD_BOX_YELLOW ? PGD_F_ON_YELLOW
                               : b[j].kind == PGD_BOX_BROKEN ? PGD_F_ON_BROKEN : PGD_F_CRASH_SIDEWALK;
          if (fl & bit) continue;
          if (obb_overlap(car, obb_of(b[j]))) fl |= bit;
        }
      }
    }
  return fl;
}
template <class MV>
DEV unsigned state_check(const MV& mv, const Grp& g, const Obb& car) { return group_or(state_check_part(mv, g.sub, g.SUB, car), g); }

// What the later phases need from the agent's route position (Navigation.current_ref_lanes / next_ref_lanes,
// navigation.py:155-183): looked up once per step after the checkpoint update, then reused by the side distances, the
// reward and the observation instead of re-walking spawn record -> road table each time.
struct RouteCtx {   // per-step by-products of an agent's after_step, consumed by reward_done (the route itself -- current / next
                    // road, reference lanes, lane counts, block id -- lives in the vehicle record: route_refresh)
  float drive;     // driving_reward * (long_now - long_last) * lateral_factor * positive_road of the step (pgdrive_env.py:209-258)
  float positive;  // +1 / -1: the sign the reference gives the speed reward on a negative road
  int clear;       // the car's box lies inside the line-free strip of its (straight) lane: no line / sidewalk contact possible
  int lane_road;   // road of the vehicle's lane after the step (the next step's trigger test reads it from here)
};

// BaseVehicle.after_step (base_vehicle.py:255-290).  `with_state_check` = false lets the caller run the line / sidewalk
// test wave-cooperatively afterwards (k_step with one env per wave) and OR the result into vflags.
// AHEAD: read the lane records of the step before the localisation (16 more registers live across it: kernels with one env per wave)
// `sp`: the slot's spawn record in memory (route arrays); `sv`: where its scalar fields are read from (the same record, or the
// caller's register copy of its head)
template <bool AHEAD = false, class MV>
DEV void after_step_vehicle(const pgd_config& cfg, const MV& mv, const Grp& g, const pgd_spawn& sp, const pgd_spawn& sv, Veh& r,
                            bool is_agent, bool with_state_check, RouteCtx& ctx) {
  float lon_v, lat_v;
  // the two lane records every agent's after_step reads -- its lane of the previous step (almost always still its lane) and the
  // first reference lane -- go out together, before the localisation: one memory round trip where there were three in a row
  const int lane0 = r.lane, first0 = (int)r.cur_first;
  const pgd_lane PL = mv.lanes[lane0];
  pgd_lane L0;
  if (AHEAD && is_agent) L0 = mv.lanes[first0];
  update_localization(mv, g, sp, r, lon_v, lat_v, PL);
  if (is_agent) {
    ctx = RouteCtx{0.0f, 1.0f, 0, -1};
    pgd_lane VL = PL;
    if (!AHEAD || r.lane != lane0) VL = mv.lanes[r.lane];
    ctx.lane_road = VL.road;
    if (!AHEAD || (int)r.cur_first != first0) L0 = mv.lanes[r.cur_first];  // AHEAD: only when a checkpoint was passed
    {
      // line / sidewalk contacts (base_vehicle.py:615-644) need no grid walk while the car's box stays inside the strip of
      // its straight lane that no such box reaches (`ex` of the device lane copy, pgd_upload_maps)
      if (VL.dir == 0.0f && VL.ex > 0.0f) {
        const float ca = fabsf(r.hx * VL.bx + r.hy * VL.by), sa = fabsf(r.hy * VL.bx - r.hx * VL.by);
        const float hl = 0.5f * sv.length, hw = 0.5f * sv.width;
        const float e_lat = hw * ca + hl * sa, e_lon = hl * ca + hw * sa;
        ctx.clear = (fabsf(lat_v) + e_lat <= VL.ex && lon_v - e_lon >= 0.0f && lon_v + e_lon <= VL.length) ? 1 : 0;
      }
    }
    unsigned fl = (unsigned)r.vflags;
    fl &= ~(PGD_F_ON_WHITE | PGD_F_ON_YELLOW | PGD_F_ON_BROKEN | PGD_F_CRASH_SIDEWALK | PGD_F_OUT_OF_ROUTE);
    if (with_state_check && !ctx.clear) fl |= state_check(mv, g, Obb{r.x, r.y, r.hx, r.hy, 0.5f * sv.length, 0.5f * sv.width});
    float lon, lat;
    lane_local(L0, r.x, r.y, lon, lat);
    float w = mv.lane_width();
    r.dl = lat + w * 0.5f;
    float range = w * (float)r.cur_n;
    if (r.blk == 'y' || r.blk == 'Y') {
      // Navigation.get_current_lateral_range on Merge / Split blocks (navigation.py:306-320,346-362): a 50 m ray from the
      // left edge of the leftmost reference lane across the road against the continuous lane lines
      float sx, sy;
      lane_position(L0, lon, -0.5f * L0.width, sx, sy);
      range = 50.0f * ray_grid(mv, sx, sy, -L0.by * 50.0f, L0.bx * 50.0f, (1u << PGD_BOX_WHITE) | (1u << PGD_BOX_YELLOW));
    }
    r.dr = range - r.dl;
    {
      // the driving term of the reward (pgdrive_env.py:209-232): longitudinal progress on the vehicle's own lane when that
      // lane belongs to the current reference road, else on the first reference lane.  Both coordinate pairs of the new
      // position were just evaluated, so the term is formed here; reward_done adds the speed term and the terminal cases.
      const bool in_ref = VL.road == (int)r.road_cur;
      float l0, t0;
      lane_local(in_ref ? VL : L0, r.lastx, r.lasty, l0, t0);
      const float l1 = in_ref ? lon_v : lon, t1 = in_ref ? lat_v : lat;
      ctx.positive = (in_ref || (cfg.marl_flags & PGD_MA_PLAIN_REWARD)) ? 1.0f : (mv.roads()[VL.road].negative ? -1.0f : 1.0f);
      const float lateral_factor = cfg.use_lateral ? clipf(1.0f - 2.0f * fabsf(t1) / w, 0.0f, 1.0f) : 1.0f;
      ctx.drive = cfg.driving_reward * (l1 - l0) * lateral_factor * ctx.positive;
    }
    if (r.dr < 0.0f || r.dl < 0.0f) fl |= PGD_F_OUT_OF_ROUTE;
    r.vflags = (int)fl;
    float dist = norm2(r.lastx - r.x, r.lasty - r.y) / 1000.0f;
    r.energy += 3.25f * expf(0.01f * speed_kmh(r.v)) * dist / 100.0f * 1000.0f;
    PHASE_MARK(19);  // after_step: side distances
  }
}

// the same test with the whole wave on one car: the (<= 2x2) grid cells under the car are flattened into one index range
template <class MV>
DEV unsigned state_check_wave(const MV& mv, const Obb& car) {
  const int lane = threadIdx.x;
  float ex = fabsf(car.ux) * car.hl + fabsf(car.uy) * car.hw, ey = fabsf(car.uy) * car.hl + fabsf(car.ux) * car.hw;
  int cx0 = max((int)floorf((car.cx - ex - mv.ox()) / mv.cell()), 0), cx1 = min((int)floorf((car.cx + ex - mv.ox()) / mv.cell()), mv.gx() - 1);
  int cy0 = max((int)floorf((car.cy - ey - mv.oy()) / mv.cell()), 0), cy1 = min((int)floorf((car.cy + ey - mv.oy()) / mv.cell()), mv.gy() - 1);
  unsigned fl = 0;
  for (int cyb = cy0; cyb <= cy1; cyb += 2)
    for (int cxb = cx0; cxb <= cx1; cxb += 2) {  // blocks of up to 2x2 cells (a car spans at most 2 cells per axis)
      int k0[4], pre[5];
      pre[0] = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int cx = cxb + (q & 1), cy = cyb + (q >> 1);
        bool in = cx <= cx1 && cy <= cy1;
        int cell = in ? cy * mv.gx() + cx : 0;
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

#endif
