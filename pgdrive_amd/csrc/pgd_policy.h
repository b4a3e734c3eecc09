// pgd_policy.h -- the policy network of the closed loop as ONE launch: actions = MLP(observation rows).
// Part of the single translation unit pgd_engine.hip.
//
// What it mirrors: pgdrive/examples/ppo_expert/numpy_expert.py:25-44 (`expert(obs)`): a three-layer tanh MLP
//   x = tanh(obs @ fc_1/kernel + fc_1/bias); x = tanh(x @ fc_2/kernel + fc_2/bias); out = x @ fc_out/kernel + fc_out/bias
// with 256 hidden units, the action = the first two outputs (the mean of the Gaussian head) -- the policy the reference's own
// closed-loop tests drive the env with (tests/test_functionality/test_expert_performance.py:48-85) and the shape of a PPO / SAC
// rollout policy in general.  As torch ops the three layers are six dependent launches of GEMMs far too small to fill the chip
// (4096 x 274 x 256): 35 us per step next to a 17 us env step (profiles/r06_notes.md).  Here: one workgroup per 16 observation rows,
// the rows and both hidden activations stay in LDS, the two 256-wide layers run on the f32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact f32, a k-ordered fma chain -- the result is what a per-thread fmaf loop gives), the 2-wide head on
// the vector ALU.  This is the one GEMM-shaped piece of the closed loop, hence the one place MFMA is used; the step itself stays
// branch / latency bound scalar geometry.
//
// Layout: weights row-major [in][out] exactly as numpy_expert.py's `kernel` arrays (x @ kernel), fp32, on the device.
// Work split: 4 waves per workgroup; wave w owns hidden columns [64 w, 64 w + 64) as FOUR 16 x 16 accumulator tiles whose columns
// interleave: tile t = columns { 64 w + 4 n + t : n = 0 .. 15 }.  Any 16 columns make a tile; with this choice the four weights a lane
// needs for one k-step -- one per tile -- are four CONSECUTIVE floats of a row of W: one 16-byte read feeds four matrix
// instructions.  (First version, round 6: tiles of 16 consecutive columns, one 4-byte read per tile and k-step: 2,176 read
// instructions per CU and launch, 21.4 us for 4096 rows -- a third of the matrix pipe's rate, bound by the issue of its reads.)
//   A fragment (activations, LDS):  lane l -> row l & 15, k = k0 + (l >> 4)
//   B fragment (weights, global, L2-resident: 545 KB for 274-256-256-2):  lane l -> k = k0 + (l >> 4), columns c0 + 4 (l & 15) + t
//   C / D of tile t: register i of lane l -> row 4 (l >> 4) + i, column c0 + 4 (l & 15) + t
// LDS row strides are = 2 (mod 32) words: the 32 lanes of a half-wave (16 rows x 2 k) then hit 32 different banks.
#ifndef PGD_POLICY_H
#define PGD_POLICY_H

#define MLP_ROWS 16
#define MLP_H 256
#define MLP_WAVES 4
#define MLP_HS (MLP_H + 2)  // 258 = 2 (mod 32)
typedef float mlp_f32x4 __attribute__((ext_vector_type(4)));

DEV_HOST int mlp_x_stride(int in_dim) {
  const int kp = (in_dim + 3) & ~3;
  return kp + ((2 - kp) % 32 + 32) % 32;
}
DEV_HOST size_t mlp_lds_bytes(int in_dim) {  // X tile | H1 | H2 | the head's weights [2][256]
  return sizeof(float) * ((size_t)MLP_ROWS * ((size_t)mlp_x_stride(in_dim) + 2 * MLP_HS) + 2 * MLP_H);
}

// one 256-wide layer for the wave's four column tiles: acc[t] += A[16 x K] (LDS, row stride a_ld) * W[K x 256] (global), K = kp (a
// multiple of four; rows of W at or beyond k_real contribute zero)
DEV void mlp_layer(const float* __restrict__ A, const int a_ld, const float* __restrict__ W, const int kp, const int k_real, const int lane,
                   const int c0, mlp_f32x4 (&acc)[4]) {
  const int arow = lane & 15, kq = lane >> 4;
  const float* ap = A + arow * a_ld + kq;
  const float4* wp = reinterpret_cast<const float4*>(W + (size_t)kq * MLP_H + c0 + 4 * (lane & 15));  // (16-byte aligned: c0, 4 n, 256 k)
  // Eight k-steps (32 rows of W) per round: eight 16-byte weight reads and eight LDS reads go out together, then 32 matrix
  // instructions (1,024 cycles of the matrix pipe).  Two register sets, ping and pong: the reads of round r + 1 are issued BEFORE the
  // matrix instructions of round r (one wave per SIMD: nothing else hides an L2 latency).
  constexpr int U = 8;
  const int k_last = k_real - 1 - kq;  // (k_real >= 4: the caller checks in_dim)
  auto load = [&](const int k0, float (&a)[U], float4 (&b)[U]) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      // rows at or beyond k_real (the padding of the last round): the reads go to the last real row instead -- unconditional reads
      // keep the round one basic block, so that the wait in front of a round's matrix instructions counts only its own reads --
      // and the activation is replaced by zero (a finite weight times zero)
      const int k = k0 + 4 * j;
      const bool in = k + kq < k_real;
      const int kc = in ? k : k_last;
      const float av = ap[kc];
      a[j] = in ? av : 0.0f;
      b[j] = wp[(size_t)kc * (MLP_H / 4)];
    }
  };
  auto mma = [&](const float (&a)[U], const float4 (&b)[U]) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j].y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j].z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j].w, acc[3], 0, 0, 0);
    }
  };
  float aP[U], aQ[U];
  float4 bP[U], bQ[U];
  // (every read is unconditional and the loop body is straight-line -- two rounds, ping then pong --, so that the wait in front of a
  // round's matrix instructions leaves the reads of the round after it in flight; an odd last round runs behind the loop: its reads
  // were the loop's last; the partial last round multiplies zeros for the k-steps past the end)
  const int rounds = (kp + 4 * U - 1) / (4 * U), pairs = rounds >> 1;
  load(0, aP, bP);
  for (int p = 0; p < pairs; ++p) {
    const int k0 = p * 8 * U;
    // (scheduling barriers: left to itself the instruction scheduler sinks the reads of the next round in between the matrix
    // instructions of this one and the wait pass then drains the read queue -- vmcnt(0) -- in the middle of a round)
    load(k0 + 4 * U, aQ, bQ);
    __builtin_amdgcn_sched_barrier(0);
    mma(aP, bP);
    __builtin_amdgcn_sched_barrier(0);
    load(k0 + 8 * U, aP, bP);
    __builtin_amdgcn_sched_barrier(0);
    mma(aQ, bQ);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (rounds & 1) mma(aP, bP);
}

// tanh in fp32 through one exponential: tanh(x) = 1 - 2 / (e^{2x} + 1), |error| of a few ulp for |x| < 9, saturating beyond
DEV float mlp_tanh(const float x) {
  const float e = __expf(2.0f * fminf(fmaxf(x, -10.0f), 10.0f));
  return 1.0f - 2.0f / (e + 1.0f);
}

// bias + tanh of the wave's four tiles into the next layer's LDS tile: a lane holds four consecutive columns of each of its four rows
DEV void mlp_store_hidden(float* __restrict__ H, const float* __restrict__ bias, const int lane, const int c0, const mlp_f32x4 (&acc)[4]) {
  const int col = c0 + 4 * (lane & 15), r4 = (lane >> 4) * 4;
  const float4 bb = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float* h = H + (r4 + i) * MLP_HS + col;  // (8-byte aligned: the row stride is even)
    reinterpret_cast<float2*>(h)[0] = make_float2(mlp_tanh(acc[0][i] + bb.x), mlp_tanh(acc[1][i] + bb.y));
    reinterpret_cast<float2*>(h)[1] = make_float2(mlp_tanh(acc[2][i] + bb.z), mlp_tanh(acc[3][i] + bb.w));
  }
}

// rows [row0, row0 + n_rows) of `obs` (row stride obs_stride floats, in_dim used) -> act[row][0..1]
template <bool TANH_OUT>
__global__ __launch_bounds__(WAVE * MLP_WAVES, 4) void k_mlp_policy(const float* __restrict__ obs, const int row0, const int n_rows,
                                                                 const int obs_stride, const int in_dim, const float* __restrict__ W1,
                                                                 const float* __restrict__ b1, const float* __restrict__ W2,
                                                                 const float* __restrict__ b2, const float* __restrict__ W3,
                                                                 const float* __restrict__ b3, const int out_cols, float* __restrict__ act) {
  extern __shared__ float mlp_lds[];
  const int kp = (in_dim + 3) & ~3, xs = mlp_x_stride(in_dim);
  float* X = mlp_lds;
  float* H1 = X + MLP_ROWS * xs;
  float* H2 = H1 + MLP_ROWS * MLP_HS;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r0 = (int)blockIdx.x * MLP_ROWS;  // (relative to row0)
  // the head's weights (256 x 2 of W3, its first two columns) go to LDS with the observation rows: read in the head itself, 32 strided
  // reads per lane behind the last barrier were 2 us of a launch that has nothing else to do by then
  float* W3s = H2 + MLP_ROWS * MLP_HS;  // [2][256]
  float w3v[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + q * WAVE * MLP_WAVES;  // 0 .. 511: (k = idx >> 1, o = idx & 1)
    w3v[q] = W3[(size_t)(idx >> 1) * out_cols + (idx & 1)];
  }
  // the 16 observation rows: whole rows, coalesced (a row is contiguous in memory); padding columns and rows past the end read zero.
  // Every read of the wave's four rows goes out before the first LDS store (rows of up to 320 floats: five chunks of 64 per row) --
  // as a read-then-store loop the prologue was twenty memory round trips in a row, a third of the launch (round 6)
  constexpr int XCH = 5;
  if (kp <= WAVE * XCH) {
    float v[MLP_ROWS / MLP_WAVES][XCH];
#pragma unroll
    for (int i = 0; i < MLP_ROWS / MLP_WAVES; ++i) {
      const int r = wave + i * MLP_WAVES;
      const bool row_in = r0 + r < n_rows;
      const float* src = obs + (size_t)(row0 + r0 + (row_in ? r : 0)) * obs_stride;
#pragma unroll
      for (int j = 0; j < XCH; ++j) {
        const int k = lane + WAVE * j;
        v[i][j] = src[k < in_dim ? k : in_dim - 1];
        if (!(row_in && k < in_dim)) v[i][j] = 0.0f;
      }
    }
#pragma unroll
    for (int i = 0; i < MLP_ROWS / MLP_WAVES; ++i)
#pragma unroll
      for (int j = 0; j < XCH; ++j) {
        const int k = lane + WAVE * j;
        if (k < kp) X[(wave + i * MLP_WAVES) * xs + k] = v[i][j];
      }
  } else
  for (int r = wave; r < MLP_ROWS; r += MLP_WAVES) {
    const bool row_in = r0 + r < n_rows;
    const float* src = obs + (size_t)(row0 + r0 + (row_in ? r : 0)) * obs_stride;
    for (int k = lane; k < kp; k += WAVE) X[r * xs + k] = (row_in && k < in_dim) ? src[k] : 0.0f;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + q * WAVE * MLP_WAVES;
    W3s[(idx & 1) * MLP_H + (idx >> 1)] = w3v[q];
  }
  __syncthreads();
  const int c0 = wave * 64;
  mlp_f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = mlp_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  mlp_layer(X, xs, W1, kp, in_dim, lane, c0, acc);
  mlp_store_hidden(H1, b1, lane, c0, acc);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = mlp_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  mlp_layer(H1, MLP_HS, W2, MLP_H, MLP_H, lane, c0, acc);
  mlp_store_hidden(H2, b2, lane, c0, acc);
  __syncthreads();
  // the head: 16 rows x 2 outputs = 32 dot products of 256, eight lanes each
  {
    const int dot = tid >> 3, part = tid & 7, r = dot >> 1, o = dot & 1;
    float s = 0.0f;
#pragma unroll 4
    for (int k = part; k < MLP_H; k += 8) s = fmaf(H2[r * MLP_HS + k], W3s[o * MLP_H + k], s);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 1);
    if (part == 0 && r0 + r < n_rows) {
      const float v = s + b3[o];
      act[(size_t)(row0 + r0 + r) * 2 + o] = TANH_OUT ? mlp_tanh(v) : v;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same network on the bf16 matrix cores with SPLIT operands (pgd_mlp_prepare + pgd_mlp_policy_prepared; `precision 1`).
// Every f32 value is carried as hi + lo, two bf16 numbers (hi = the value rounded to bf16, lo = the rounded remainder: 16 bits of
// mantissa together), and a product a * b is formed as a_hi b_hi + a_hi b_lo + a_lo b_hi on `v_mfma_f32_16x16x32_bf16` (f32
// accumulation; the dropped a_lo b_lo is 2^-16 of the product).  Three matrix instructions of 16 cycles cover K = 32 where the exact
// f32 form needs eight of 32 cycles: the matrix pipe's share of the launch drops from 7.7 to 1.5 us, and the launch is bound by what
// streams the weights through the CU's L1 (same bytes as f32: 2 x 2 instead of 4 per value).  Error against float64: ~3e-5 on an
// action in [-1, 1] (test_mlp_policy_matches_the_numpy_expert holds 1e-4); the exact-f32 kernel above stays the default.
// Weights are PREPARED once per policy update (k_mlp_prepare): split, and re-laid out so that a lane's B fragment of a tile and a
// 32-row chunk -- eight values of one column -- is one 16-byte read:  [layer][chunk c][wave w][tile t][plane hi / lo][lane] x 16 B,
// lane l = (column 64 w + 16 t + (l & 15), rows 32 c + 8 (l >> 4) .. + 7); behind the two layers: b1, b2, the head's 2 x 256
// weights and b3 as f32.  (Which K index the hardware gives element j of lane group g does not matter: A and B fragments put the
// SAME k into the same (g, j) slot.)
typedef __bf16 mlp_bf16x8 __attribute__((ext_vector_type(8)));
DEV unsigned mlp_bf16_rne(const float x) {  // round to nearest even; finite inputs
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
DEV float mlp_bf16_f(const unsigned b) { return __uint_as_float(b << 16); }
DEV void mlp_split(const float x, unsigned& hi, unsigned& lo) {
  hi = mlp_bf16_rne(x);
  lo = mlp_bf16_rne(x - mlp_bf16_f(hi));
}
DEV_HOST int mlp_chunks(int k) { return (k + 31) / 32; }
DEV_HOST size_t mlp_prepared_layer_bytes(int k) { return (size_t)mlp_chunks(k) * MLP_WAVES * 4 * 2 * WAVE * 16; }
DEV_HOST size_t mlp_prepared_bytes(int in_dim) {
  return mlp_prepared_layer_bytes(in_dim) + mlp_prepared_layer_bytes(MLP_H) + sizeof(float) * (2 * MLP_H + 2 * MLP_H + 4);
}
DEV_HOST int mlp_bf_stride(int kp) { return kp + 8; }  // bf16 elements per LDS row: (kp / 2 + 4) words = 4 (odd) -> 16 rows x 16 B hit 64 banks
DEV_HOST size_t mlp_bf_lds_bytes(int in_dim) {
  return 2 * (size_t)MLP_ROWS * (2 * (size_t)mlp_bf_stride(32 * mlp_chunks(in_dim)) + 4 * (size_t)mlp_bf_stride(MLP_H)) + sizeof(float) * 2 * MLP_H;
}

// one thread per (layer, chunk, wave, tile, lane): eight weights of one column, split and packed
__global__ __launch_bounds__(256) void k_mlp_prepare(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                                                     const float* __restrict__ b2, const float* __restrict__ W3, const float* __restrict__ b3,
                                                     const int in_dim, const int out_cols, uint4* __restrict__ prep) {
  const int c1 = mlp_chunks(in_dim), c2 = mlp_chunks(MLP_H);
  const int per_chunk = MLP_WAVES * 4 * WAVE;
  const int n_frag = (c1 + c2) * per_chunk;
  const int id = (int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x;
  if (id < n_frag) {
    const bool second = id >= c1 * per_chunk;
    const int q = second ? id - c1 * per_chunk : id;
    const int c = q / per_chunk, r = q - c * per_chunk, w = r / (4 * WAVE), t = (r / WAVE) & 3, l = r & (WAVE - 1);
    const float* W = second ? W2 : W1;
    const int kreal = second ? MLP_H : in_dim;
    const int col = 64 * w + 16 * t + (l & 15), k0 = 32 * c + 8 * (l >> 4);
    unsigned h[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) mlp_split(k0 + j < kreal ? W[(size_t)(k0 + j) * MLP_H + col] : 0.0f, h[j], lo[j]);
    uint4* dst = prep + (second ? mlp_prepared_layer_bytes(in_dim) / 16 : 0) + ((size_t)((c * MLP_WAVES + w) * 4 + t) * 2) * WAVE + l;
    dst[0] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    dst[WAVE] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
  }
  float* tail = reinterpret_cast<float*>(reinterpret_cast<char*>(prep) + mlp_prepared_layer_bytes(in_dim) + mlp_prepared_layer_bytes(MLP_H));
  if (id < MLP_H) {
    tail[id] = b1[id];
    tail[MLP_H + id] = b2[id];
    tail[2 * MLP_H + id] = W3[(size_t)id * out_cols];          // head, output 0
    tail[3 * MLP_H + id] = W3[(size_t)id * out_cols + 1];      // head, output 1
  }
  if (id < 2) tail[4 * MLP_H + id] = b3[id];
}

// one 256-wide layer, split operands: A planes in LDS (row stride a_ld bf16 elements), B fragments from the prepared buffer
DEV void mlp_layer_bf(const unsigned short* __restrict__ AH, const unsigned short* __restrict__ AL, const int a_ld, const uint4* __restrict__ Bp,
                      const int chunks, const int lane, const int wave, mlp_f32x4 (&acc)[4]) {
  const int arow = lane & 15, g = lane >> 4;
  const uint4* ah = reinterpret_cast<const uint4*>(AH + arow * a_ld + 8 * g);  // (16-byte aligned: a_ld and the chunk offsets are multiples of 8)
  const uint4* al = reinterpret_cast<const uint4*>(AL + arow * a_ld + 8 * g);
  const uint4* bp = Bp + (size_t)wave * 4 * 2 * WAVE + lane;
  auto load = [&](const int c, uint4 (&a)[2], uint4 (&b)[8]) {
    const int cc = c < chunks ? c : chunks - 1;  // (a read past the last chunk repeats it: unconditional reads, never used)
    a[0] = ah[cc * 4];  // 32 bf16 = 4 x 16 B per chunk
    a[1] = al[cc * 4];
#pragma unroll
    for (int q = 0; q < 8; ++q) b[q] = bp[((size_t)cc * MLP_WAVES * 4 * 2 + q) * WAVE];  // q = 2 t + plane
  };
  auto mma = [&](const uint4 (&a)[2], const uint4 (&b)[8]) {
    const mlp_bf16x8 aH = __builtin_bit_cast(mlp_bf16x8, a[0]), aL = __builtin_bit_cast(mlp_bf16x8, a[1]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const mlp_bf16x8 bH = __builtin_bit_cast(mlp_bf16x8, b[2 * t]), bL = __builtin_bit_cast(mlp_bf16x8, b[2 * t + 1]);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aL, bH, acc[t], 0, 0, 0);  // (the small terms first)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aH, bL, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aH, bH, acc[t], 0, 0, 0);
    }
  };
  uint4 aP[2], aQ[2], bP[8], bQ[8];
  const int pairs = chunks >> 1;
  load(0, aP, bP);
  for (int p = 0; p < pairs; ++p) {
    load(2 * p + 1, aQ, bQ);
    __builtin_amdgcn_sched_barrier(0);
    mma(aP, bP);
    __builtin_amdgcn_sched_barrier(0);
    load(2 * p + 2, aP, bP);
    __builtin_amdgcn_sched_barrier(0);
    mma(aQ, bQ);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (chunks & 1) mma(aP, bP);
}

DEV void mlp_store_hidden_bf(unsigned short* __restrict__ HH, unsigned short* __restrict__ HL, const float* __restrict__ bias, const int lane,
                             const int wave, const mlp_f32x4 (&acc)[4]) {
  const int n = lane & 15, r4 = (lane >> 4) * 4, hs = mlp_bf_stride(MLP_H);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = 64 * wave + 16 * t + n;
    const float bb = bias[col];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned hi, lo;
      mlp_split(mlp_tanh(acc[t][i] + bb), hi, lo);
      HH[(r4 + i) * hs + col] = (unsigned short)hi;
      HL[(r4 + i) * hs + col] = (unsigned short)lo;
    }
  }
}

template <bool TANH_OUT>
__global__ __launch_bounds__(WAVE * MLP_WAVES, 4) void k_mlp_policy_bf(const float* __restrict__ obs, const int row0, const int n_rows,
                                                                       const int obs_stride, const int in_dim, const uint4* __restrict__ prep,
                                                                       float* __restrict__ act) {
  extern __shared__ float mlp_lds[];
  const int c1 = mlp_chunks(in_dim), kp = 32 * c1, xs = mlp_bf_stride(kp), hs = mlp_bf_stride(MLP_H);
  unsigned short* XH = reinterpret_cast<unsigned short*>(mlp_lds);
  unsigned short* XL = XH + MLP_ROWS * xs;
  unsigned short* H1H = XL + MLP_ROWS * xs;
  unsigned short* H1L = H1H + MLP_ROWS * hs;
  unsigned short* H2H = H1L + MLP_ROWS * hs;
  unsigned short* H2L = H2H + MLP_ROWS * hs;
  float* W3s = reinterpret_cast<float*>(H2L + MLP_ROWS * hs);  // [2][256] (the stride terms are multiples of 8 elements: 4-byte aligned)
  const float* tail = reinterpret_cast<const float*>(reinterpret_cast<const char*>(prep) + mlp_prepared_layer_bytes(in_dim) + mlp_prepared_layer_bytes(MLP_H));
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r0 = (int)blockIdx.x * MLP_ROWS;
  float w3v[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) w3v[q] = tail[2 * MLP_H + tid + q * WAVE * MLP_WAVES];
  // the observation rows, split on their way into LDS (all reads of the wave's four rows before the first store; rows of up to 320 floats)
  constexpr int XCH = 5;
  for (int i = 0; i < MLP_ROWS / MLP_WAVES; ++i) {
    const int r = wave + i * MLP_WAVES;
    const bool row_in = r0 + r < n_rows;
    const float* src = obs + (size_t)(row0 + r0 + (row_in ? r : 0)) * obs_stride;
    if (kp <= WAVE * XCH) {
      float v[XCH];
#pragma unroll
      for (int j = 0; j < XCH; ++j) {
        const int k = lane + WAVE * j;
        v[j] = src[k < in_dim ? k : in_dim - 1];
        if (!(row_in && k < in_dim)) v[j] = 0.0f;
      }
#pragma unroll
      for (int j = 0; j < XCH; ++j) {
        const int k = lane + WAVE * j;
        if (k < kp) { unsigned hi, lo; mlp_split(v[j], hi, lo); XH[r * xs + k] = (unsigned short)hi; XL[r * xs + k] = (unsigned short)lo; }
      }
    } else {
      for (int k = lane; k < kp; k += WAVE) {
        unsigned hi, lo;
        mlp_split((row_in && k < in_dim) ? src[k] : 0.0f, hi, lo);
        XH[r * xs + k] = (unsigned short)hi; XL[r * xs + k] = (unsigned short)lo;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) W3s[tid + q * WAVE * MLP_WAVES] = w3v[q];
  __syncthreads();
  mlp_f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = mlp_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  mlp_layer_bf(XH, XL, xs, prep, c1, lane, wave, acc);
  mlp_store_hidden_bf(H1H, H1L, tail, lane, wave, acc);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = mlp_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  mlp_layer_bf(H1H, H1L, hs, prep + mlp_prepared_layer_bytes(in_dim) / 16, mlp_chunks(MLP_H), lane, wave, acc);
  mlp_store_hidden_bf(H2H, H2L, tail + MLP_H, lane, wave, acc);
  __syncthreads();
  {
    const int dot = tid >> 3, part = tid & 7, r = dot >> 1, o = dot & 1;
    float s = 0.0f;
#pragma unroll 4
    for (int k = part; k < MLP_H; k += 8) s = fmaf(mlp_bf16_f(H2H[r * hs + k]) + mlp_bf16_f(H2L[r * hs + k]), W3s[o * MLP_H + k], s);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 1);
    if (part == 0 && r0 + r < n_rows) {
      const float v = s + tail[4 * MLP_H + o];
      act[(size_t)(row0 + r0 + r) * 2 + o] = TANH_OUT ? mlp_tanh(v) : v;
    }
  }
}

#endif
