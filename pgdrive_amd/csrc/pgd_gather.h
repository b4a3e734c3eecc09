// pgd_gather.h -- the per-step gather of (obs, reward, done) rows by DIRECT PEER WRITES (pgd_gather_* of pgdrive_hip.h).
// Part of the single translation unit pgd_engine.hip (included at its end).
//
// xGMI is point to point (7 links per GPU): a ring all-gather moves 7/8 of the result over ONE link per GPU, a direct
// exchange uses all seven at once.  Every rank owns nbuf receive buffers [world * n_rows][row_floats]
// and a small control area in one hipMalloc block, exports it over HIP IPC and maps the blocks of its peers.  Per step:
//   k_step (pgd_step_packed) writes the rank's rows straight into its own slice of its own receive buffer;
//   k_peer_push copies that slice into the same slice of every peer's buffer (plain stores over the links), then raises
//     flags[buf][rank] = seq on the peer (system-scope release);
//   k_peer_wait (consumer side) spins until flags[buf][p] >= seq for every peer p (system-scope loads);
//   k_peer_release tells every peer that this rank has finished reading buffer `buf` (acks[buf][rank] = seq on the peer);
//     a sender does not overwrite a peer's buffer before that ack (checked at the start of k_peer_push).
// Sequence numbers either come with the call (seq > 0) or -- seq == 0 -- from a per-buffer counter on the device that the release
// call advances (buf + 1, then + nbuf per use: what a host that counts steps would pass).  With device-side sequences the calls of a
// step are the same every time, so  wait, release, pgd_step_packed, push  of nbuf (or any multiple of nbuf) consecutive steps can be
// captured in ONE HIP graph and replayed: the host then costs one graph launch per cycle instead of four calls per step.
// Spins are bounded: a peer that never arrives raises the error word instead of hanging the queue.
#ifndef PGD_GATHER_H
#define PGD_GATHER_H

#define PGD_GATHER_MAX_WORLD 64
#define PGD_GATHER_SPIN_LIMIT (1u << 24)  // x ~1 us sleep: about 15-20 s

struct GatherCtl {  // lives at the end of every rank's block; written by remote kernels
  int flags[4][PGD_GATHER_MAX_WORLD];  // [buf][sender]: highest sequence number whose rows have landed here
  int acks[4][PGD_GATHER_MAX_WORLD];   // [buf][reader]: highest sequence number the reader has released (written INTO the sender's block)
  int err;                             // != 0: a bounded spin ran out
  int pad[63];
};

struct pgd_gather {
  int device, world, rank, n_rows, row_floats, nbuf;
  size_t buf_bytes, ctl_off, total_bytes;
  char* base;                              // own block
  char* peer_base[PGD_GATHER_MAX_WORLD];   // mapped blocks (own entry = base)
  bool connected[PGD_GATHER_MAX_WORLD];
  int fine;                                // 1: the block is fine-grained memory, 0: the runtime refused it (or PGD_GATHER_COARSE) and it is plain hipMalloc
  int* counters;                           // [nbuf][world] block-arrival counters of k_peer_push (own, device)
  int* dseq;                               // [4] device-side sequence of each buffer (seq == 0 calls): its last push, 0 = never used
  char** d_peer_base;                      // device copy of peer_base
};

DEV int sys_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
DEV void sys_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// grid = (blocks_per_peer, world); block (x, p) copies chunk x of the own slice into peer p's buffer
__global__ __launch_bounds__(256) void k_peer_push(char* const* __restrict__ peer_base, char* base, int world, int rank, int buf, int nbuf,
                                                    int seq_arg, size_t buf_bytes, size_t slice_bytes, size_t ctl_off, int* counters,
                                                    const int* __restrict__ dseq) {
  const int seq = seq_arg > 0 ? seq_arg : dseq[buf];
  const int p = blockIdx.y;
  if (p == rank || peer_base[p] == nullptr) return;
  GatherCtl* my_ctl = reinterpret_cast<GatherCtl*>(base + ctl_off);
  // flow control: peer p must have released what this buffer held before (sequence seq - nbuf)
  if (seq > nbuf) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
      unsigned spins = 0;
      while (sys_load(&my_ctl->acks[buf][p]) < seq - nbuf && spins < PGD_GATHER_SPIN_LIMIT) { __builtin_amdgcn_s_sleep(32); ++spins; }
      ok = spins < PGD_GATHER_SPIN_LIMIT;
      if (!ok) sys_store(&my_ctl->err, 1);
    }
    __syncthreads();
    if (!ok) return;
  }
  const size_t off = (size_t)buf * buf_bytes + (size_t)rank * slice_bytes;
  const uint4* src = reinterpret_cast<const uint4*>(base + off);
  uint4* dst = reinterpret_cast<uint4*>(peer_base[p] + off);
  const size_t n16 = slice_bytes / 16;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (size_t)gridDim.x * blockDim.x) dst[k] = src[k];
  __threadfence_system();  // the rows are on their way before the arrival is counted
  __syncthreads();
  if (threadIdx.x == 0) {
    const int arrived = atomicAdd(&counters[buf * world + p], 1) + 1;
    if (arrived == (int)gridDim.x) {  // last block for this peer: every block's rows are released
      counters[buf * world + p] = 0;
      __threadfence_system();
      GatherCtl* peer_ctl = reinterpret_cast<GatherCtl*>(peer_base[p] + ctl_off);
      sys_store(&peer_ctl->flags[buf][rank], seq);
    }
  }
}

__global__ void k_peer_wait(char* base, size_t ctl_off, int world, int rank, int buf, int seq_arg, const int* __restrict__ dseq) {
  const int seq = seq_arg > 0 ? seq_arg : dseq[buf];  // (0: the buffer has never been pushed: nothing to wait for)
  GatherCtl* ctl = reinterpret_cast<GatherCtl*>(base + ctl_off);
  const int p = threadIdx.x;
  if (p >= world || p == rank) return;
  unsigned spins = 0;
  while (sys_load(&ctl->flags[buf][p]) < seq && spins < PGD_GATHER_SPIN_LIMIT) { __builtin_amdgcn_s_sleep(32); ++spins; }
  if (spins >= PGD_GATHER_SPIN_LIMIT) sys_store(&ctl->err, 2);
  __threadfence_system();  // acquire: the rows are read by later kernels of this stream
}

__global__ void k_peer_release(char* const* __restrict__ peer_base, size_t ctl_off, int world, int rank, int buf, int seq_arg, int nbuf,
                               int* dseq) {
  const int p = threadIdx.x;
  const int seq = seq_arg > 0 ? seq_arg : dseq[buf];
  if (seq > 0 && p < world && p != rank && peer_base[p] != nullptr) {
    GatherCtl* peer_ctl = reinterpret_cast<GatherCtl*>(peer_base[p] + ctl_off);
    sys_store(&peer_ctl->acks[buf][rank], seq);
  }
  if (seq_arg <= 0) {  // device-side sequences: this buffer's next use is its next generation
    __syncthreads();
    if (threadIdx.x == 0) dseq[buf] = seq == 0 ? buf + 1 : seq + nbuf;
  }
}

extern "C" {

int pgd_gather_create(int device, int world, int rank, int n_rows, int row_floats, int nbuf, pgd_gather_handle* out) {
  if (!out || world < 1 || world > PGD_GATHER_MAX_WORLD || rank < 0 || rank >= world || n_rows <= 0 || row_floats <= 0 || nbuf < 1 ||
      nbuf > 4)
    return PGD_ERR_ARG;
  if (((size_t)n_rows * row_floats * 4) % 16 != 0) return PGD_ERR_ARG;  // slices are copied in 16-byte units
  HIPCHK(hipSetDevice(device));
  pgd_gather* g = (pgd_gather*)calloc(1, sizeof(pgd_gather));
  g->device = device; g->world = world; g->rank = rank; g->n_rows = n_rows; g->row_floats = row_floats; g->nbuf = nbuf;
  g->buf_bytes = (size_t)world * n_rows * row_floats * 4;
  g->ctl_off = ((size_t)nbuf * g->buf_bytes + 255) & ~(size_t)255;
  g->total_bytes = g->ctl_off + sizeof(GatherCtl);
  // Peers write the rows and the flag / ack words of this block over xGMI while kernels of this device poll and read them:
  // fine-grained (device-coherent across agents) memory, as RCCL uses for its own buffers -- ordinary coarse-grained hipMalloc
  // memory is only guaranteed coherent at kernel boundaries, so a polling kernel could keep seeing a stale flag line.
  // PGD_GATHER_COARSE=1 forces plain hipMalloc (A/B; reported by pgd_gather_mem_kind, bench.py prints `gather_mem`).
  // A runtime that refuses the flag makes pgd_gather_create FAIL (PGD_ERR_HIP): polling kernels on coarse memory may hang or read
  // stale rows, and a caller that can fall back has a better fallback than that (pgdrive_amd/dist.py: the RCCL transports).  Only
  // the explicit switch runs the protocol on coarse memory (the fallback used to be silent but for `gather_mem`).
  g->fine = 1;
  if (getenv("PGD_GATHER_COARSE")) {
    g->fine = 0;
    HIPCHK(hipMalloc((void**)&g->base, g->total_bytes));
  } else if (hipExtMallocWithFlags((void**)&g->base, g->total_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    fprintf(stderr, "pgd_gather_create: the runtime refused %zu bytes of fine-grained memory; the peer transport needs it "
                    "(PGD_GATHER_COARSE=1 forces coarse-grained memory, at the caller's risk)\n", g->total_bytes);
    free(g);
    return PGD_ERR_HIP;
  }
  HIPCHK(hipMemset(g->base, 0, g->total_bytes));
  HIPCHK(hipMalloc((void**)&g->counters, sizeof(int) * 4 * PGD_GATHER_MAX_WORLD));
  HIPCHK(hipMemset(g->counters, 0, sizeof(int) * 4 * PGD_GATHER_MAX_WORLD));
  HIPCHK(hipMalloc((void**)&g->dseq, sizeof(int) * 4));
  HIPCHK(hipMemset(g->dseq, 0, sizeof(int) * 4));
  HIPCHK(hipMalloc((void**)&g->d_peer_base, sizeof(char*) * PGD_GATHER_MAX_WORLD));
  g->peer_base[rank] = g->base;
  g->connected[rank] = true;
  HIPCHK(hipMemcpy(g->d_peer_base, g->peer_base, sizeof(char*) * PGD_GATHER_MAX_WORLD, hipMemcpyHostToDevice));
  HIPCHK(hipDeviceSynchronize());
  *out = g;
  return PGD_OK;
}

int pgd_gather_buffer(pgd_gather_handle g, int buf, float** d_recv) {
  if (!g || !d_recv || buf < 0 || buf >= g->nbuf) return PGD_ERR_ARG;
  *d_recv = reinterpret_cast<float*>(g->base + (size_t)buf * g->buf_bytes);
  return PGD_OK;
}

int pgd_gather_export(pgd_gather_handle g, void* handle_bytes) {
  if (!g || !handle_bytes) return PGD_ERR_ARG;
  static_assert(sizeof(hipIpcMemHandle_t) <= PGD_GATHER_HANDLE_BYTES, "IPC handle does not fit the ABI's handle blob");
  hipIpcMemHandle_t hd;
  HIPCHK(hipSetDevice(g->device));
  HIPCHK(hipIpcGetMemHandle(&hd, g->base));
  memset(handle_bytes, 0, PGD_GATHER_HANDLE_BYTES);
  memcpy(handle_bytes, &hd, sizeof(hd));
  return PGD_OK;
}

int pgd_gather_connect(pgd_gather_handle g, int peer, const void* handle_bytes) {
  if (!g || !handle_bytes || peer < 0 || peer >= g->world || peer == g->rank || g->connected[peer]) return PGD_ERR_ARG;
  hipIpcMemHandle_t hd;
  memcpy(&hd, handle_bytes, sizeof(hd));
  void* p = nullptr;
  HIPCHK(hipSetDevice(g->device));
  HIPCHK(hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess));
  g->peer_base[peer] = (char*)p;
  g->connected[peer] = true;
  HIPCHK(hipMemcpy(g->d_peer_base, g->peer_base, sizeof(char*) * PGD_GATHER_MAX_WORLD, hipMemcpyHostToDevice));
  return PGD_OK;
}

int pgd_gather_push(pgd_gather_handle g, int buf, int seq, void* hip_stream) {
  if (!g || buf < 0 || buf >= g->nbuf || seq < 0) return PGD_ERR_ARG;
  if (g->world == 1) return PGD_OK;
  const size_t slice = (size_t)g->n_rows * g->row_floats * 4;
  int bpp = (int)((slice / 16 + 256 * 8 - 1) / (256 * 8));  // ~8 x 16 B per thread
  bpp = bpp < 1 ? 1 : (bpp > 64 ? 64 : bpp);
  hipLaunchKernelGGL(k_peer_push, dim3(bpp, g->world), dim3(256), 0, (hipStream_t)hip_stream, g->d_peer_base, g->base, g->world,
                     g->rank, buf, g->nbuf, seq, g->buf_bytes, slice, g->ctl_off, g->counters, g->dseq);
  HIPCHK(hipGetLastError());
  return PGD_OK;
}

int pgd_gather_wait(pgd_gather_handle g, int buf, int seq, void* hip_stream) {
  if (!g || buf < 0 || buf >= g->nbuf || seq < 0) return PGD_ERR_ARG;
  if (g->world == 1) return PGD_OK;
  hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(PGD_GATHER_MAX_WORLD), 0, (hipStream_t)hip_stream, g->base, g->ctl_off, g->world, g->rank,
                     buf, seq, g->dseq);
  HIPCHK(hipGetLastError());
  return PGD_OK;
}

int pgd_gather_release(pgd_gather_handle g, int buf, int seq, void* hip_stream) {
  if (!g || buf < 0 || buf >= g->nbuf || seq < 0) return PGD_ERR_ARG;
  if (g->world == 1) return PGD_OK;
  hipLaunchKernelGGL(k_peer_release, dim3(1), dim3(PGD_GATHER_MAX_WORLD), 0, (hipStream_t)hip_stream, g->d_peer_base, g->ctl_off,
                     g->world, g->rank, buf, seq, g->nbuf, g->dseq);
  HIPCHK(hipGetLastError());
  return PGD_OK;
}

int pgd_gather_status(pgd_gather_handle g, int* err) {
  if (!g || !err) return PGD_ERR_ARG;
  HIPCHK(hipSetDevice(g->device));
  HIPCHK(hipMemcpy(err, g->base + g->ctl_off + offsetof(GatherCtl, err), sizeof(int), hipMemcpyDeviceToHost));
  return PGD_OK;
}

int pgd_gather_mem_kind(pgd_gather_handle g, int* fine_grained) {
  if (!g || !fine_grained) return PGD_ERR_ARG;
  *fine_grained = g->fine;
  return PGD_OK;
}

int pgd_gather_destroy(pgd_gather_handle g) {
  if (!g) return PGD_ERR_ARG;
  (void)hipSetDevice(g->device);
  (void)hipDeviceSynchronize();
  for (int p = 0; p < g->world; ++p)
    if (p != g->rank && g->peer_base[p]) (void)hipIpcCloseMemHandle(g->peer_base[p]);
  (void)hipFree(g->base);
  (void)hipFree(g->counters);
  (void)hipFree(g->dseq);
  (void)hipFree(g->d_peer_base);
  free(g);
  return PGD_OK;
}

}  // extern "C"

#endif
