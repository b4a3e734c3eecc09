// pgd_topdown.h -- top-down (bird's-eye) multi-channel observation: obs/top_down_obs_multi_channel.py:18-280 as a rasteriser
// kernel.  Part of the single translation unit pgd_engine.hip (included at its end, before pgd_gather.h).
//
// The reference draws the map and the vehicles with pygame on 2000 x 2000 canvases, crops a window around the ego, rotates it
// so that the ego heads up (ObservationWindow, top_down_obs_impl.py:15-90), converts to grey and stacks channels
//   [road_network * 2, past_pos, traffic_flow(t), traffic_flow(t - skip), ...]      (top_down_obs_multi_channel.py:203-247)
// into a [R, R, 2 + frame_stack] image in [0, 1] (TopDownPGDriveEnv: R = 84, distance = 30 m, frame_stack 3, post_stack 5,
// frame_skip 5; envs/top_down_env.py:8-42).  pygame is not available here (nor on the GPU box), so pixel parity with its
// polygon / line / rotozoom / smoothscale rasterisation is UNPINNED; this kernel evaluates the same scene analytically, one
// pixel centre at a time, and the oracle restates exactly this definition in fp64:
//   window   pixel (row i, col j) <-> ego frame: forward = (R/2 - i - 0.5) / s, right = (j + 0.5 - R/2) / s, s = R / (2 distance)
//            px per metre (the crop of ObservationWindow: +-distance around the ego, ego heading up, top_down_obs_impl.py:30-82)
//   ch 0     road_network: like the reference, the map is drawn ONCE -- a raster per scenario at TD_TEXEL = 0.25 m (the
//            reference's background canvas: 2000 px over the map's extent + 20 m, i.e. 3-4 px / m), every texel classified
//            analytically at its centre (k_topdown_raster) -- and the window samples it.  Single RGB frame: the nearest texel.
//            Multi-channel: the reference renders this channel at 2 R and halves it with pygame.transform.smoothscale
//            (top_down_obs_multi_channel.py:214-217: for an exact halving an area average), after rotozoom has resampled the canvas
//            bilinearly -- anti-aliased greys on the edges of lines and route lanes.  Here (round 5): the raster is PRE-AVERAGED at
//            upload -- a half-resolution raster whose 0.5 m cell holds how many of its 2 x 2 texels are line / route lane
//            (k_topdown_reduce) -- and a pixel takes the cell under its centre: value = (n_line * 35 + n_navi * 64) * 2 / 255 / 4,
//            ONE gather per pixel (round 4 averaged four point samples a quarter pixel around the centre: four gathers, 138 -> 200 us;
//            neither form is pygame's pixel -- that stays unpinned -- and both are the same analytic scene, area-averaged).  Classes:
//            2 * 35/255 where the texel centre lies within 0.25 m (half of LANE_LINE_WIDTH = 0.5, or half a window pixel if
//            that is more) of a lane-line box of the map (continuous and broken lines; LANE_LINE_COLOR (35,35,35)); else
//            2 * 64/255 inside a lane (|lateral| <= width / 2, 0 <= longitudinal <= length) of a road on the ego's route
//            (draw_navigation with (64,64,64)); else 0.  Deviation: stripes follow the map's physical broken-line boxes, not
//            the renderer's cosmetic 3 m / 5 m pattern.
//   ch 1     past_pos: 1 at the pixels of the ego's positions t, t - skip, ... (post_stack entries at most), in the CURRENT ego
//            frame, with the reference's own scale R / distance (twice the window's: reproduced as is) and its clip
//   ch 2..   traffic_flow at t, t - skip, ...: grey(100,200,255)/255 = 0.6917 inside the box of every other vehicle (all
//            BaseVehicles of the engine: waiting and driving traffic, broken-down vehicles; not cones / barriers), each
//            frame in the ego frame OF ITS OWN TIME; headings within 2 deg of 0 snap to 0 (top_down_obs_multi_channel.py:141-150)
// State: the engine keeps, per env, the last (post_stack - 1) * skip + 1 ego positions and the last (frame_stack - 1) * skip + 1
// pose sets (every slot + the ego) and re-rasterises old frames from poses (3 KB per env instead of 78 KB of stored images).
// After a reset the history is the first frame repeated (the reference fills its deque the same way; its past_pos deque shows
// the previous episode's points in the first frame after a reset -- not reproduced).
#ifndef PGD_TOPDOWN_H
#define PGD_TOPDOWN_H
#ifndef TD_CHUNK
#define TD_CHUNK 7424  /* pixels per gather / write chunk of k_topdown: whole image rows, a multiple of 8 of them (84 x 84 fits in one) */
#endif
// The rasters are stored in tiles of 8 x 8 texels = one 64-byte line each: the window is rotated against the raster, and a
// tile of 8 x 8 window pixels (the lanes of a gathering wave) then touches a dozen lines instead of one line per lane
DEV long long td_tiled(int ix, int iy, int tiles_x) { return ((long long)(iy >> 3) * tiles_x + (ix >> 3)) * 64 + ((iy & 7) << 3) + (ix & 7); }

struct TopDown {
  int R, C, frame_stack, post_stack, frame_skip, n_pos, n_frames;
  float distance;
  float2* pos;    // [N][n_pos]   newest first
  float4* pose;   // [N][n_frames][V]  (x, y, hx, hy); hx = hy = 0: not drawn; newest first; slot 0 = the ego of that time
  int* n_hist;    // [N] valid entries of pos
  // road-network raster, one per scenario (the route is the scenario's): texel = TD_TEXEL m, value 0 / 1 (route lane) / 2 (line)
  const uint8_t* tex;        // all rasters back to back
  const long long* tex_off;  // [n_scen] offset of the scenario's raster
  const uint8_t* occ;        // per 64-byte line of `tex` (an 8 x 8 tile of texels / cells): 1 = something is drawn in it (null: not built)
  float line_r;
  int rgb;  // pgd_topdown_config.mode == 1: one RGB frame (C = 3): lines grey, the ego green, the others blue; no route, no history
};
#define TD_RGB_LINE (35.0f / 255.0f)
#define TD_TEXEL 0.25f

#define TD_LINE 0.27450980392156865f   // 2 * 35 / 255
#define TD_NAVI 0.50196078431372548f   // 2 * 64 / 255
#define TD_VEH 0.69164705882352939f    // (0.299 * 100 + 0.587 * 200 + 0.114 * 255) / 255

// road-network class of a world point: 2 = within line_r of a lane-line box, 1 = inside a lane of a road of the route, 0 = neither
DEV int td_classify(const MapView& mv, const uint32_t* route, float wx, float wy, float line_r) {
  const pgd_map& m = *mv.m;
  const int cx = (int)floorf((wx - m.ox) / m.cell), cy = (int)floorf((wy - m.oy) / m.cell);
  bool line = false, navi = false;
  // a line box within line_r of the point may be registered in a neighbouring cell only: look at the 3 x 3 block when the
  // point is closer than line_r to a cell border, else at its own cell
  const float fx = (wx - m.ox) - (float)cx * m.cell, fy = (wy - m.oy) - (float)cy * m.cell;
  const int x0 = fx < line_r ? cx - 1 : cx, x1 = fx > m.cell - line_r ? cx + 1 : cx;
  const int y0 = fy < line_r ? cy - 1 : cy, y1 = fy > m.cell - line_r ? cy + 1 : cy;
  for (int yy = y0; yy <= y1 && !line; ++yy)
    for (int xx = x0; xx <= x1 && !line; ++xx) {
      if (xx < 0 || yy < 0 || xx >= m.gx || yy >= m.gy) continue;
      const int cell = yy * m.gx + xx;
      const int k1 = cell_first(mv.cstart[cell + 1]);
      if (xx == cx && yy == cy && !navi)
        for (int k = cell_first(mv.cstart[cell]); k < cell_mid(mv.cstart[cell]); ++k) {  // lane boxes: candidates for the exact lane test
          const pgd_box b = mv.cbox[k];
          const pgd_lane& L = mv.lanes[b.lane];
          if (!((route[L.road >> 5] >> (L.road & 31)) & 1u)) continue;
          float lo, la;
          lane_local(L, wx, wy, lo, la);
          if (fabsf(la) <= 0.5f * L.width && lo >= 0.0f && lo <= L.length) { navi = true; break; }
        }
      for (int k = cell_mid(mv.cstart[cell]); k < k1; ++k) {
        const pgd_box b = mv.cbox[k];
        if (b.kind == PGD_BOX_SIDEWALK) continue;
        if (point_obb_dist(obb_of(b), wx, wy) <= line_r) { line = true; break; }
      }
    }
  return line ? 2 : (navi ? 1 : 0);
}

// The road-network raster of one scenario (once per upload: what the reference's 2000 x 2000 background canvas is, here at a fixed
// TD_TEXEL = 0.25 m): every texel classified analytically at its centre.  k_topdown samples it with the nearest texel.
__global__ __launch_bounds__(256) void k_topdown_raster(PgdDev d, int scen, float line_r, uint8_t* __restrict__ tex, int W, int H) {
  __shared__ uint32_t s_route[128];
  for (int k = threadIdx.x; k < 128; k += 256) s_route[k] = 0u;
  __syncthreads();
  const pgd_spawn& sp = d.spawns[(size_t)scen * d.sstride];
  if ((int)threadIdx.x < sp.n_ckpt - 1) { const int r = sp.ckpt_road[threadIdx.x]; if (r >= 0) atomicOr(&s_route[r >> 5], 1u << (r & 31)); }
  __syncthreads();
  const MapView mv = map_view_of(d, d.scen_map + scen);
  const int tbw = (W + 7) >> 3, tbh = (H + 7) >> 3;
  const long long n = (long long)tbw * tbh * 64;  // tiled layout (td_tiled); texels past the edge of the last tiles are 0
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
    const long long tile = p >> 6;
    const int in_t = (int)(p & 63), by = (int)(tile / tbw), bx = (int)(tile - (long long)by * tbw);
    const int iy = by * 8 + (in_t >> 3), ix = bx * 8 + (in_t & 7);
    tex[p] = (ix < W && iy < H) ? (uint8_t)td_classify(mv, s_route, mv.m->ox + ((float)ix + 0.5f) * TD_TEXEL, mv.m->oy + ((float)iy + 0.5f) * TD_TEXEL, line_r)
                                : (uint8_t)0;
  }
}

// Half-resolution raster of the multi-channel road channel: cell (x, y) covers the texels (2x .. 2x+1, 2y .. 2y+1) of the 0.25 m
// raster; its byte = number of line texels | number of route-lane texels << 4 (texels beyond the raster count as nothing).
__global__ __launch_bounds__(256) void k_topdown_reduce(const uint8_t* __restrict__ fine, uint8_t* __restrict__ coarse, int W, int H) {
  const int W2 = (W + 1) >> 1, H2 = (H + 1) >> 1;
  const int tbw = (W + 7) >> 3, tbw2 = (W2 + 7) >> 3, tbh2 = (H2 + 7) >> 3;
  const long long n = (long long)tbw2 * tbh2 * 64;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
    const long long tile = p >> 6;
    const int in_t = (int)(p & 63), by = (int)(tile / tbw2), bx = (int)(tile - (long long)by * tbw2);
    const int y2 = by * 8 + (in_t >> 3), x2 = bx * 8 + (in_t & 7);
    int acc = 0;
    if (x2 < W2 && y2 < H2)
      for (int q = 0; q < 4; ++q) {
        const int x = 2 * x2 + (q & 1), y = 2 * y2 + (q >> 1);
        if (x < W && y < H) { const int c = fine[td_tiled(x, y, tbw)]; acc += c == 2 ? 1 : (c == 1 ? 16 : 0); }
      }
    coarse[p] = (uint8_t)acc;
  }
}

// One byte per 64-byte line of the rasters (a tile of 8 x 8 texels, or of 8 x 8 half-resolution cells): is anything drawn in it.  Two
// thirds of the 8 x 8 pixel tiles of a window see no road at all; k_topdown asks these bytes before it computes 64 raster addresses.
__global__ __launch_bounds__(256) void k_topdown_occ(const uint8_t* __restrict__ tex, uint8_t* __restrict__ occ, long long n_lines) {
  for (long long l = (long long)blockIdx.x * 256 + threadIdx.x; l < n_lines; l += (long long)gridDim.x * 256) {
    const uint4* q = reinterpret_cast<const uint4*>(tex + l * 64);
    unsigned any = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint4 v = q[k]; any |= v.x | v.y | v.z | v.w; }
    occ[l] = any ? 1 : 0;
  }
}

#ifndef TD_OCC
#define TD_OCC 1
#endif
#ifndef TD_OCC_U8
#define TD_OCC_U8 6  /* the byte kernel: six waves per SIMD (80 registers, five of them spilled) against five without the bound: 77.4 -> 71.5 us; seven: 70.7 us with twenty spilled */
#endif
#ifndef TD_NK
#define TD_NK 4
#endif
// RGB: the single RGB frame (TopDown::rgb) as a compile-time fact: the multi-channel kernel carries none of its branches
// U8: the image as bytes in [0, 255] -- the reference's `rgb_clip=False` (top_down_obs_multi_channel.py:208-211, 253-256, 277-280: the
// uint8 pygame values themselves instead of float32 / 255): a quarter of the bytes of a kernel that is bound by its writes.  The byte
// of a pixel is the float image's value x 255, truncated like numpy's astype(uint8): (lines x 35 + route texels x 64) / 2 on the
// road channel, 255 on the past positions, 176 in a vehicle box; 35 / (50, 200, 0) / (100, 200, 255) in the RGB frame.
template <bool RGB, bool U8 = false>
__global__ __launch_bounds__(256, (U8 ? TD_OCC_U8 : TD_OCC)) void k_topdown(PgdDev d, TopDown t_in, uint8_t* __restrict__ fill, void* __restrict__ img_v) {
  using PX = typename std::conditional<U8, uint8_t, float>::type;
  TopDown t = t_in;
  t.rgb = RGB ? 1 : 0;
  // dynamic LDS, sized by the engine's V instead of the 64-slot maximum (more blocks per CU: a block alternates between phases
  // that wait for reads and a phase that only writes, and the CU overlaps them across blocks):
  //   pose history after this step's insertion [n_frames][V] | per stacked frame the visible boxes [4][V] (centre, axis) | [4][V] half extents
  extern __shared__ float4 s_dyn4[];
  __shared__ float2 s_pos[64];               // ego position history (n_pos <= 64)
  __shared__ float s_hl[MAXV], s_hw[MAXV];
  __shared__ int s_nhist;
  // per stacked frame: the vehicles that can show up in the window, already in the ego frame of that time
  __shared__ int s_nvis[4];
  __shared__ __attribute__((aligned(16))) uint8_t s_cls[TD_CHUNK];  // texel classes of a chunk of pixels: gathered first, then the chunk is written out
  __shared__ unsigned short s_tl[256];  // the band's 8 x 8 pixel tiles that can see something of the raster: row of tiles << 8 | column
  __shared__ int s_ntl;
  __shared__ float s_out[256 * 8];   // one batch of 256 pixels x C channels, written out linearly (coalesced)
  const int e = blockIdx.x, tid = threadIdx.x, V = d.V;
  float4* s_pose = s_dyn4;                                  // [f * V + s]
  float4* s_vc = s_dyn4 + (size_t)t.n_frames * V;           // [f * V + k]: (forward, right) of the box centre, (forward, right) of its long axis
  float2* s_vh = reinterpret_cast<float2*>(s_vc + 4 * V);   // [f * V + k]: half extents
  const RecPiece* recs = rec_block(d.rec, (size_t)e, V);
  const int scen = d.ei[(size_t)e * PGD_NEI + EI_SCEN];
  const pgd_spawn* spb = d.spawns + (size_t)scen * d.sstride;
  const MapView mv = map_view_of(d, d.scen_map + scen);
  const bool refill = fill[e] != 0;
  // ---- history: shift by one and insert the current state (or fill everything with it after a reset) ----
  float4* hp = t.pose + (size_t)e * t.n_frames * V;
  float2* pp = t.pos + (size_t)e * t.n_pos;
  for (int k = tid; k < t.n_frames * V; k += 256) {
    const int f = k / V, s = k - f * V;
    float4 q;
    if (f == 0 || refill) {
      Veh rc;  // (the first 64 bytes: pose, heading vector, spawn index, status)
      load_rec_head(recs, V, s, rc);
      const bool drawn = (rc.status == ST_PENDING || rc.status == ST_ACTIVE || rc.status == ST_DYING) && spb[rc.spawn].kind == PGD_OBJ_VEHICLE;
      float hx = rc.hx, hy = rc.hy;
      if (s != 0 && fabsf(rc.th) <= 2.0f * PGD_PI / 180.0f) { hx = 1.0f; hy = 0.0f; }  // the reference snaps small headings of the others
      q = drawn ? make_float4(rc.x, rc.y, hx, hy) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else q = hp[(size_t)(f - 1) * V + s];
    s_pose[f * V + s] = q;
  }
  for (int k = tid; k < t.n_pos; k += 256) s_pos[k] = (k == 0 || refill) ? make_float2(rec_float(recs, V, 0, RW_X), rec_float(recs, V, 0, RW_Y)) : pp[k - 1];
  if (tid < V) { const pgd_spawn& so = spb[rec_spawn(recs, V, tid)]; s_hl[tid] = 0.5f * so.length; s_hw[tid] = 0.5f * so.width; }
  if (tid == 0) s_nhist = refill ? 1 : min(t.n_hist[e] + 1, t.n_pos);
  __syncthreads();
  for (int k = tid; k < t.n_frames * V; k += 256) hp[k] = s_pose[k];
  for (int k = tid; k < t.n_pos; k += 256) pp[k] = s_pos[k];
  if (tid == 0) { t.n_hist[e] = s_nhist; fill[e] = 0; }
  __syncthreads();
  // ---- per frame: cull the vehicles against the window and move them into that frame's ego coordinates (one wave per frame) ----
  {
    const int w = tid >> 6, lane = tid & 63;
    for (int f = w; f < t.frame_stack && f < 4; f += 4) {
      const int fi = f * t.frame_skip;
      const float4 eg = s_pose[fi * V];
      bool vis = false;
      float4 vc = make_float4(0.f, 0.f, 1.f, 0.f);
      if (lane >= 1 && lane < V) {
        const float4 q = s_pose[fi * V + lane];
        if (!(q.z == 0.0f && q.w == 0.0f)) {
          const float dx = q.x - eg.x, dy = q.y - eg.y;
          vc = make_float4(dx * eg.z + dy * eg.w, dy * eg.z - dx * eg.w, q.z * eg.z + q.w * eg.w, q.w * eg.z - q.z * eg.w);
          const float reach = t.distance + s_hl[lane] + s_hw[lane];
          vis = fabsf(vc.x) <= reach && fabsf(vc.y) <= reach;
        }
      }
      const unsigned long long m = __ballot(vis);
      if (vis) {
        const int k = __popcll(m & ((1ull << lane) - 1ull));
        s_vc[f * V + k] = vc;
        s_vh[f * V + k] = make_float2(s_hl[lane], s_hw[lane]);
      }
      if (lane == 0) s_nvis[f] = __popcll(m);
    }
  }
  __syncthreads();
  // ---- rasterise ----
  const int R = t.R, C = t.C;
  const float s_px = (float)R / (2.0f * t.distance), inv_s = 1.0f / s_px;
  const pgd_map& m = *mv.m;
  const int tw = (int)((float)m.gx * m.cell / TD_TEXEL), th = (int)((float)m.gy * m.cell / TD_TEXEL);
  const uint8_t* tex = t.tex + t.tex_off[scen];
  PX* out = reinterpret_cast<PX*>(img_v) + (size_t)e * R * R * C;
  // The image is almost empty outside channel 0: the code below writes the road channel and zeros, then the few pixels the
  // vehicles cover are set.  Loads and stores share one completion counter on this hardware and complete out of order with
  // respect to each other, so a wave that waits for a texel also waits for every store it has in flight: the texel classes
  // of a chunk of pixels are therefore gathered into LDS first (many reads in flight, no store pending), and the store loop
  // that follows contains no global read at all -- each wave streams its batches of 64 consecutive pixels out of its own
  // staging area and never waits for memory or for the other three waves.
  const int wv = tid >> 6, lane = tid & 63, n_pix = R * R;
  // the wave's staging area: 64 pixels x C (<= 6) floats -- or 256 pixels x C bytes: the same 2 KB --, then written out linearly
  PX* so = reinterpret_cast<PX*>(s_out) + wv * (U8 ? 256 * 8 : 64 * 8);
  for (int k = lane; k < 64 * 8; k += 64) s_out[wv * 64 * 8 + k] = 0.0f;  // channels 1.. stay zero: only ch 0 is rewritten per batch
  // every batch is then a whole number of 16-byte stores but for the image's last one (bytes: every env's image starts on a 16-byte boundary)
  const bool vec_ok = U8 ? ((n_pix * C) & 15) == 0 : (((n_pix * C) & 3) == 0 && ((64 * C) & 3) == 0);
  const float4 eg0 = s_pose[0];
  const float m_ox = m.ox, m_oy = m.oy;
  const int tbw = (tw + 7) >> 3;
  // raster entry under the centre of pixel (pi, pj): the road network around the CURRENT ego pose (right = heading rotated by
  // +90 deg in the engine's x / y frame).  RGB frame: the nearest 0.25 m texel (its class).  Multi-channel: the 0.5 m cell of the
  // pre-averaged raster that holds it (line / route-lane counts of the cell's 2 x 2 texels, k_topdown_reduce) -- the cell index is
  // the texel index halved, and a centre whose texel lies beyond the raster reads nothing
  const int tbw2 = (((tw + 1) >> 1) + 7) >> 3;
  auto texel_addr = [&](int pi, int pj, bool on, bool& in) -> long long {
    const float fw = ((float)R * 0.5f - (float)pi - 0.5f) * inv_s, rg = ((float)pj + 0.5f - (float)R * 0.5f) * inv_s;
    const float wx = eg0.x + fw * eg0.z - rg * eg0.w, wy = eg0.y + fw * eg0.w + rg * eg0.z;
    const int ix = (int)floorf((wx - m_ox) * (1.0f / TD_TEXEL)), iy = (int)floorf((wy - m_oy) * (1.0f / TD_TEXEL));
    in = on && ix >= 0 && iy >= 0 && ix < tw && iy < th;
    return in ? (RGB ? td_tiled(ix, iy, tbw) : td_tiled(ix >> 1, iy >> 1, tbw2)) : 0ll;
  };
  const int rows_c = min((TD_CHUNK / R) & ~7, (R + 7) & ~7);  // image rows per chunk
  for (int row0 = 0; row0 < R; row0 += rows_c) {
    const int row1 = min(row0 + rows_c, R);
    const int c0 = row0 * R, c1 = row1 * R;
    // ---- which tiles: an 8 x 8 pixel tile of the window is gathered only if the bounding box of its pixel centres (the window map is
    // affine: the box of the four corner centres, a texel of slack on every side) touches a raster line in which something is drawn
    // (TopDown::occ); the classes of the other tiles are zero.  One tile per thread.
    const int txn = (R + 7) >> 3, n_tiles = ((row1 - row0 + 7) >> 3) * txn;
    {
      const uint8_t* occ = t.occ ? t.occ + (t.tex_off[scen] >> 6) : nullptr;
      const int lw = RGB ? tbw : tbw2, lh = RGB ? ((th + 7) >> 3) : ((((th + 1) >> 1) + 7) >> 3);  // raster lines per row / rows of lines
      if (tid == 0) s_ntl = 0;
      for (int k = tid; k < ((c1 - c0 + 3) >> 2); k += 256) reinterpret_cast<uint32_t*>(s_cls)[k] = 0u;
      __syncthreads();
      for (int q = tid; q < n_tiles; q += 256) {
        const int ty = q / txn, tx = q - ty * txn;
        bool take = true;
        if (occ) {
          const int i0 = row0 + ty * 8, i1 = min(i0 + 7, row1 - 1), j0 = tx * 8, j1 = min(j0 + 7, R - 1);
          float u0 = 3.0e38f, u1 = -3.0e38f, v0 = 3.0e38f, v1 = -3.0e38f;
#pragma unroll
          for (int cnr = 0; cnr < 4; ++cnr) {
            const int pi = (cnr & 1) ? i1 : i0, pj = (cnr & 2) ? j1 : j0;
            const float fw = ((float)R * 0.5f - (float)pi - 0.5f) * inv_s, rg = ((float)pj + 0.5f - (float)R * 0.5f) * inv_s;
            const float wx = eg0.x + fw * eg0.z - rg * eg0.w, wy = eg0.y + fw * eg0.w + rg * eg0.z;
            const float u = (wx - m_ox) * (1.0f / TD_TEXEL), v = (wy - m_oy) * (1.0f / TD_TEXEL);
            u0 = fminf(u0, u); u1 = fmaxf(u1, u); v0 = fminf(v0, v); v1 = fmaxf(v1, v);
          }
          const int sh = RGB ? 3 : 4;  // texel index -> line index (multi-channel: cells of two texels, lines of eight cells)
          const int lx0 = max((int)floorf(u0 - 1.0f) >> sh, 0), lx1 = min((int)floorf(u1 + 1.0f) >> sh, lw - 1);
          const int ly0 = max((int)floorf(v0 - 1.0f) >> sh, 0), ly1 = min((int)floorf(v1 + 1.0f) >> sh, lh - 1);
          int any = 0;
          for (int ly = ly0; ly <= ly1; ++ly)
            for (int lx = lx0; lx <= lx1; ++lx) any |= (int)occ[(long long)ly * lw + lx];
          take = any != 0;
        }
        if (take) s_tl[atomicAdd(&s_ntl, 1)] = (unsigned short)((ty << 8) | tx);
      }
      __syncthreads();
    }
    // ---- gather: the waves take the listed tiles in turn, lane = pixel of the tile, 4 tiles in flight (80 VGPRs: 6 blocks per CU; 8 in flight cost a block)
    {
      const int n_tl = s_ntl;
      constexpr int NK = TD_NK;
      // (the tile's row and column come out of the list as scalars: as `q / txn` per lane and tile the division was a third of the
      // loop's vector instructions, and the loop is bound by them: 45 of the byte image's 89 us)
      const int wv_s = __builtin_amdgcn_readfirstlane(wv);
      for (int n0 = wv_s; n0 < n_tl; n0 += 4 * NK) {
        int v[NK], pix[NK];
#pragma unroll
        for (int u = 0; u < NK; ++u) {
          const int n = n0 + 4 * u;
          const int code = __builtin_amdgcn_readfirstlane((int)s_tl[min(n, n_tl - 1)]);
          const int ty = code >> 8, tx = code & 255;
          const int i = row0 + ty * 8 + (lane >> 3), j = tx * 8 + (lane & 7);
          const bool on = n < n_tl && i < row1 && j < R;
          bool in;
          v[u] = tex[texel_addr(i, j, on, in)];  // RGB: the class; multi-channel: low nibble = line texels of the cell, high nibble = route-lane texels
          if (!in) v[u] = 0;
          pix[u] = on ? i * R + j - c0 : -1;
        }
#pragma unroll
        for (int u = 0; u < NK; ++u)
          if (pix[u] >= 0) s_cls[pix[u]] = (uint8_t)v[u];
      }
    }
    __syncthreads();
    // ---- stream out: no global read in this loop
    auto store_loop = [&](auto vec_tag) {
      constexpr bool VEC = decltype(vec_tag)::value;
      for (int p0 = c0 + wv * 64; p0 < c1; p0 += 256) {
        const int n_here = min(64, c1 - p0);
        const int cls = lane < n_here ? (int)s_cls[p0 - c0 + lane] : 0;
        if (U8) {
          if (t.rgb) { const PX g = (PX)(cls == 2 ? 35 : 0); so[lane * 3] = g; so[lane * 3 + 1] = g; so[lane * 3 + 2] = g; }
          else so[lane * C] = (PX)(((cls & 15) * 35 + (cls >> 4) * 64) >> 1);
        } else
        if (t.rgb) { const float g = cls == 2 ? TD_RGB_LINE : 0.0f; so[lane * 3] = (PX)g; so[lane * 3 + 1] = (PX)g; so[lane * 3 + 2] = (PX)g; }
        else so[lane * C] = (PX)((float)(cls & 15) * (0.25f * TD_LINE) + (float)(cls >> 4) * (0.25f * TD_NAVI));
        row_sync<true>();  // the wave's own LDS traffic only
        const int nf = n_here * C;
        PX* dst = out + (size_t)p0 * C;
        if (U8) {  // 64 x C bytes = at most 32 sixteen-byte pieces: one predicated store, the tail of the image's last batch byte by byte
          const int n16 = VEC ? nf >> 4 : 0;
          if (lane < n16) OBS_ST(reinterpret_cast<obs_f4*>(dst) + lane, reinterpret_cast<const obs_f4*>(so)[lane]);
  #pragma unroll 1
        for (int k = (n16 << 4) + lane; k < nf; k += 64) dst[k] = so[k];
        } else
        if (VEC) {  // 64 x C floats = at most 128 float4 (C <= 8): two predicated stores, no loop
          const int n4 = nf >> 2;
          if (lane < n4) OBS_ST(reinterpret_cast<obs_f4*>(dst) + lane, reinterpret_cast<const obs_f4*>(so)[lane]);
          if (lane + 64 < n4) OBS_ST(reinterpret_cast<obs_f4*>(dst) + lane + 64, reinterpret_cast<const obs_f4*>(so)[lane + 64]);
        } else {
          for (int k = lane; k < nf; k += 64) dst[k] = so[k];
        }
        row_sync<true>();
      }
    };
    // bytes: 64 pixels are 320 bytes, twenty lanes' worth of a store -- the rounds above were what the byte image's stream-out cost
    // (27 of 89 us).  A wave takes 256 consecutive pixels per round instead: four classes per lane into the staging area, then up to 96
    // sixteen-byte pieces in two stores per lane; a quarter of the rounds and of their LDS round trips.
    if (U8 && vec_ok && (rows_c >= R || ((rows_c * R * C) & 15) == 0)) {  // (every band starts on a 16-byte boundary)
      for (int p0 = c0 + wv * 256; p0 < c1; p0 += 1024) {
        const int n_here = min(256, c1 - p0);
#pragma unroll 1
        for (int u = 0; u < 4; ++u) {
          const int k = u * 64 + lane;
          const int cls = k < n_here ? (int)s_cls[p0 - c0 + k] : 0;
          if (RGB) { const PX g = (PX)(cls == 2 ? 35 : 0); so[k * 3] = g; so[k * 3 + 1] = g; so[k * 3 + 2] = g; }
          else so[k * C] = (PX)(((cls & 15) * 35 + (cls >> 4) * 64) >> 1);
        }
        row_sync<true>();
        const int nf = n_here * C, n16 = nf >> 4;
        PX* dst = out + (size_t)p0 * C;
        if (lane < n16) OBS_ST(reinterpret_cast<obs_f4*>(dst) + lane, reinterpret_cast<const obs_f4*>(so)[lane]);
        if (lane + 64 < n16) OBS_ST(reinterpret_cast<obs_f4*>(dst) + lane + 64, reinterpret_cast<const obs_f4*>(so)[lane + 64]);
#pragma unroll 1
        for (int k = (n16 << 4) + lane; k < nf; k += 64) dst[k] = so[k];
        row_sync<true>();
      }
    } else
    if (vec_ok) store_loop(std::true_type{});
    else store_loop(std::false_type{});
    __syncthreads();
  }
  __syncthreads();
  // ch 2..: the other vehicles at t, t - skip, ...: the pixels whose centre (fwd, rgt) lies inside a box, in that frame's ego
  // coordinates.  One wave per (frame, box) pair in turn; its lanes walk the pixel rectangle around the box (circumradius + a
  // pixel of slack), 8 x 8 at a time, with the exact point-in-box test.
  // (RGB frame: the ego first -- VehicleGraphics.GREEN, its heading snapped below 2 degrees like everybody's while the window
  // turns with the true one -- then the others over it, VehicleGraphics.BLUE; obs/top_down_obs.py:150-166)
  auto paint = [&](PX* px, int f) {
    if (U8) {
      if (!t.rgb) { px[2 + f] = (PX)176; return; }  // (int)(TD_VEH * 255) = (int)176.37
      px[0] = (PX)(f < 0 ? 50 : 100); px[1] = (PX)200; px[2] = (PX)(f < 0 ? 0 : 255);
      return;
    }
    if (!t.rgb) { px[2 + f] = (PX)TD_VEH; return; }
    px[0] = (PX)(f < 0 ? 50.0f / 255.0f : 100.0f / 255.0f); px[1] = (PX)(200.0f / 255.0f); px[2] = (PX)(f < 0 ? 0.0f : 1.0f);
  };
  if (t.rgb) {
    if (wv == 0 && !(eg0.z == 0.0f && eg0.w == 0.0f)) {
      const bool snap = fabsf(rec_float(recs, V, 0, RW_TH)) <= 2.0f * PGD_PI / 180.0f;
      const float ax = snap ? eg0.z : 1.0f, ay = snap ? -eg0.w : 0.0f;  // (1, 0) turned into the window's frame; unsnapped: straight up
      const float hl = s_hl[0], hw = s_hw[0];
      const float rad = (hl + hw) * s_px + 1.5f, cc = (float)R * 0.5f - 0.5f;
      const int ia = max((int)floorf(cc - rad), 0), ib = min((int)ceilf(cc + rad), R - 1);
      for (int ti = ia; ti <= ib; ti += 8)
        for (int tj = ia; tj <= ib; tj += 8) {
          const int pi = ti + (lane >> 3), pj = tj + (lane & 7);
          if (pi > ib || pj > ib) continue;
          const float fwd = ((float)R * 0.5f - (float)pi - 0.5f) * inv_s, rgt = ((float)pj + 0.5f - (float)R * 0.5f) * inv_s;
          if (fabsf(fwd * ax + rgt * ay) <= hl && fabsf(rgt * ax - fwd * ay) <= hw) paint(out + ((size_t)pi * R + pj) * C, -1);
        }
    }
    __syncthreads();
  }
  for (int q = wv; q < 4 * MAXV; q += 4) {
    const int f = q / MAXV, k = q - f * MAXV;
    if (f >= t.frame_stack || k >= s_nvis[f]) continue;
    const float4 b = s_vc[f * V + k];
    const float2 hh = s_vh[f * V + k];
    const float rad = (hh.x + hh.y) * s_px + 1.5f;
    const float ci = (float)R * 0.5f - 0.5f - b.x * s_px, cj = b.y * s_px + (float)R * 0.5f - 0.5f;
    const int ia = max((int)floorf(ci - rad), 0), ib = min((int)ceilf(ci + rad), R - 1);
    const int ja = max((int)floorf(cj - rad), 0), jb = min((int)ceilf(cj + rad), R - 1);
    for (int ti = ia; ti <= ib; ti += 8)
      for (int tj = ja; tj <= jb; tj += 8) {
        const int pi = ti + (lane >> 3), pj = tj + (lane & 7);
        if (pi > ib || pj > jb) continue;
        const float fwd = ((float)R * 0.5f - (float)pi - 0.5f) * inv_s, rgt = ((float)pj + 0.5f - (float)R * 0.5f) * inv_s;
        const float dx = fwd - b.x, dy = rgt - b.y;
        if (fabsf(dx * b.z + dy * b.w) <= hh.x && fabsf(dy * b.z - dx * b.w) <= hh.y) paint(out + ((size_t)pi * R + pj) * C, f);
      }
  }
  if (t.rgb) return;
  // ch 1: past positions of the ego, newest first, in the current ego frame (top_down_obs_multi_channel.py:152-170)
  if (tid < t.post_stack) {
    const int k = tid * t.frame_skip;
    if (k < s_nhist) {
      const float4 eg = s_pose[0];
      const bool snap = fabsf(rec_float(recs, V, 0, RW_TH)) <= 2.0f * PGD_PI / 180.0f;  // the reference rotates by the snapped ego heading here
      const float ehx = snap ? 1.0f : eg.z, ehy = snap ? 0.0f : eg.w;
      const float dx = s_pos[k].x - eg.x, dy = s_pos[k].y - eg.y, sc = (float)R / t.distance;
      float u = (dy * ehx - dx * ehy) * sc + (float)R * 0.5f, vv = -(dx * ehx + dy * ehy) * sc + (float)R * 0.5f;
      u = clipf(u, -(float)R, (float)R); vv = clipf(vv, -(float)R, (float)R);
      const int jj = (int)floorf(u), ii = (int)floorf(vv);
      if (ii >= 0 && jj >= 0 && ii < R && jj < R) out[((size_t)ii * R + jj) * C + 1] = U8 ? (PX)255 : (PX)1;
    }
  }
}

struct pgd_topdown_state {
  TopDown t;
  uint8_t* tex;          // device rasters
  long long* tex_off;    // device offsets
  uint8_t* occ;          // device: one byte per 64-byte line of `tex`
  bool tex_dirty;        // maps / scenarios were uploaded since the rasters were built
};

// (re)build the road-network rasters of every scenario (needs maps + scenarios on the device)
static int topdown_build_rasters(pgd_engine* h) {
  pgd_topdown_state* s = h->topdown;
  if (!s || !h->have_maps || !h->have_scen || !h->h_maps || !h->h_scen) return PGD_OK;
  const int n_scen = (int)h->h_scen->size();
  std::vector<long long> off((size_t)n_scen);
  long long total = 0;
  const bool rgb = s->t.rgb != 0;  // RGB frame: the 0.25 m rasters themselves; multi-channel: their pre-averaged halves
  long long fine_max = 0;
  for (int k = 0; k < n_scen; ++k) {
    const pgd_map& M = (*h->h_maps)[(size_t)(*h->h_scen)[(size_t)k].map];
    off[(size_t)k] = total;
    const long long W = (long long)((float)M.gx * M.cell / TD_TEXEL), H = (long long)((float)M.gy * M.cell / TD_TEXEL);
    const long long fine = ((W + 7) / 8) * ((H + 7) / 8) * 64;
    const long long W2 = (W + 1) / 2, H2 = (H + 1) / 2;
    fine_max = std::max(fine_max, fine);
    total += rgb ? fine : ((W2 + 7) / 8) * ((H2 + 7) / 8) * 64;
  }
  if (s->tex) { HIPCHK(hipFree(s->tex)); s->tex = nullptr; }
  if (s->tex_off) { HIPCHK(hipFree(s->tex_off)); s->tex_off = nullptr; }
  if (s->occ) { HIPCHK(hipFree(s->occ)); s->occ = nullptr; }
  {  // one raster per scenario, 16 bytes per square metre of map extent (1-2 MB per PGDrive-v0 map; a quarter of that pre-averaged):
     // a bank of a thousand scenarios is gigabytes -- refuse with a message instead of failing inside hipMalloc
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    if ((size_t)(total + fine_max) > free_b / 2) {
      fprintf(stderr, "[pgdrive_hip] top-down rasters of %d scenarios need %.1f GB (%.1f GB free): use fewer scenarios per engine\n",
              n_scen, (double)(total + fine_max) / 1e9, (double)free_b / 1e9);
      return PGD_ERR_STATE;
    }
  }
  HIPCHK(hipMalloc(&s->tex, (size_t)(total > 0 ? total : 1)));
  HIPCHK(hipMalloc(&s->tex_off, sizeof(long long) * (size_t)n_scen));
  uint8_t* fine_tmp = nullptr;  // multi-channel: one scenario's 0.25 m raster at a time, reduced into its half-resolution raster
  if (!rgb) HIPCHK(hipMalloc(&fine_tmp, (size_t)(fine_max > 0 ? fine_max : 1)));
  HIPCHK(hipMemcpyAsync(s->tex_off, off.data(), sizeof(long long) * (size_t)n_scen, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int k = 0; k < n_scen; ++k) {
    const pgd_map& M = (*h->h_maps)[(size_t)(*h->h_scen)[(size_t)k].map];
    const int W = (int)((float)M.gx * M.cell / TD_TEXEL), H = (int)((float)M.gy * M.cell / TD_TEXEL);
    const long long n = (long long)((W + 7) / 8) * ((H + 7) / 8) * 64;
    const int blocks = (int)std::min<long long>((n + 255) / 256, 8192);
    uint8_t* fine = rgb ? s->tex + off[(size_t)k] : fine_tmp;
    hipLaunchKernelGGL(k_topdown_raster, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, h->stream, h->d, k, s->t.line_r, fine, W, H);
    if (!rgb) {
      const long long n2 = (long long)(((W + 1) / 2 + 7) / 8) * (((H + 1) / 2 + 7) / 8) * 64;
      const int blocks2 = (int)std::min<long long>((n2 + 255) / 256, 8192);
      hipLaunchKernelGGL(k_topdown_reduce, dim3(blocks2 > 0 ? blocks2 : 1), dim3(256), 0, h->stream, fine, s->tex + off[(size_t)k], W, H);
    }
  }
  if (fine_tmp) { HIPCHK(hipStreamSynchronize(h->stream)); HIPCHK(hipFree(fine_tmp)); }
  s->t.occ = nullptr;
  if (total > 0 && !getenv("PGD_TD_NO_OCC")) {  // (PGD_TD_NO_OCC=1, tests and A/B: every tile of a window is gathered)
    const long long n_lines = total / 64;  // (every raster is a whole number of 64-byte lines)
    HIPCHK(hipMalloc(&s->occ, (size_t)n_lines));
    hipLaunchKernelGGL(k_topdown_occ, dim3((unsigned)std::min<long long>((n_lines + 255) / 256, 8192)), dim3(256), 0, h->stream, s->tex, s->occ, n_lines);
    s->t.occ = s->occ;
  }
  HIPCHK(hipGetLastError());
  s->t.tex = s->tex;
  s->t.tex_off = s->tex_off;
  s->tex_dirty = false;
  return PGD_OK;
}

extern "C" {

int pgd_topdown_channels(const pgd_topdown_config* c) { return c ? (c->mode == 1 ? 3 : 2 + c->frame_stack) : 0; }

int pgd_topdown_enable(pgd_handle h, const pgd_topdown_config* c) {
  if (!h || !c || c->resolution < 8 || c->resolution > 512 || !(c->distance > 0.0f) || c->frame_stack < 1 || c->post_stack < 1 ||
      c->frame_skip < 1)
    return PGD_ERR_ARG;
  if (c->frame_stack > 4) return PGD_ERR_ARG;  // one wave of the block per stacked frame; s_out holds 256 pixels x (2 + 4) channels
  if (h->d.A != 1) return PGD_ERR_ARG;  // "Don't support multi-agent top-down observation yet" (top_down_obs_multi_channel.py:130)
  if (c->mode != 0 && c->mode != 1) return PGD_ERR_ARG;
  const bool rgb = c->mode == 1;  // one frame of the present state: no history beyond it
  const int n_pos = rgb ? 1 : (c->post_stack - 1) * c->frame_skip + 1, n_frames = rgb ? 1 : (c->frame_stack - 1) * c->frame_skip + 1;
  if (n_pos > 64 || n_frames > 16) return PGD_ERR_ARG;
  HIPCHK(hipSetDevice(h->device));
  if (!h->topdown) h->topdown = (pgd_topdown_state*)calloc(1, sizeof(pgd_topdown_state));
  pgd_topdown_state* s = h->topdown;
  if (s->t.pos) { (void)hipFree(s->t.pos); (void)hipFree(s->t.pose); (void)hipFree(s->t.n_hist); }
  s->t = TopDown{c->resolution, rgb ? 3 : 2 + c->frame_stack, rgb ? 1 : c->frame_stack, rgb ? 1 : c->post_stack, rgb ? 1 : c->frame_skip,
                 n_pos, n_frames, c->distance, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0f, rgb ? 1 : 0};
  s->t.line_r = fmaxf(0.25f, 0.5f * (2.0f * c->distance) / (float)c->resolution);
  s->tex_dirty = true;
  const size_t N = (size_t)h->d.N;
  HIPCHK(hipMalloc(&s->t.pos, sizeof(float2) * N * n_pos));
  HIPCHK(hipMalloc(&s->t.pose, sizeof(float4) * N * n_frames * h->d.V));
  HIPCHK(hipMalloc(&s->t.n_hist, sizeof(int) * N));
  HIPCHK(hipMemsetAsync(s->t.n_hist, 0, sizeof(int) * N, h->stream));
  if (!h->d.bev_fill) {
    HIPCHK(hipMalloc(&h->d.bev_fill, N));
  }
  HIPCHK(hipMemsetAsync(h->d.bev_fill, 1, N, h->stream));  // every env starts with a filled history
  return PGD_OK;
}

static int observe_topdown_impl(pgd_handle h, void* d_img, bool u8) {
  if (!h || !d_img) return PGD_ERR_ARG;
  if (!h->topdown || !h->have_maps || !h->have_scen) return PGD_ERR_STATE;
  if (u8 && (reinterpret_cast<uintptr_t>(d_img) & 15u)) return PGD_ERR_ARG;  // (the byte image is written sixteen bytes at a time)
  HIPCHK(hipSetDevice(h->device));
  if (h->topdown->tex_dirty) { int rc = topdown_build_rasters(h); if (rc) return rc; }
  const size_t dyn = sizeof(float4) * ((size_t)h->topdown->t.n_frames + 4) * h->d.V + sizeof(float2) * 4 * (size_t)h->d.V;
  void (*k)(PgdDev, TopDown, uint8_t*, void*) = h->topdown->t.rgb ? (u8 ? k_topdown<true, true> : k_topdown<true, false>)
                                                                 : (u8 ? k_topdown<false, true> : k_topdown<false, false>);
  hipLaunchKernelGGL(k, dim3(h->d.N), dim3(256), dyn, h->stream, h->d, h->topdown->t, h->d.bev_fill, d_img);
  HIPCHK(hipGetLastError());
  return PGD_OK;
}
int pgd_observe_topdown(pgd_handle h, float* d_img) { return observe_topdown_impl(h, d_img, false); }
int pgd_observe_topdown_u8(pgd_handle h, uint8_t* d_img) { return observe_topdown_impl(h, d_img, true); }

}  // extern "C"

static void topdown_mark_dirty(pgd_engine* h) { if (h->topdown) h->topdown->tex_dirty = true; }

static void topdown_free(pgd_engine* h) {
  if (!h->topdown) return;
  if (h->topdown->t.pos) { (void)hipFree(h->topdown->t.pos); (void)hipFree(h->topdown->t.pose); (void)hipFree(h->topdown->t.n_hist); }
  if (h->topdown->tex) (void)hipFree(h->topdown->tex);
  if (h->topdown->tex_off) (void)hipFree(h->topdown->tex_off);
  if (h->topdown->occ) (void)hipFree(h->topdown->occ);
  if (h->d.bev_fill) (void)hipFree(h->d.bev_fill);
  free(h->topdown);
  h->topdown = nullptr;
}

#endif
