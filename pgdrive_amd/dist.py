"""Env-sharded data parallelism: one process per GPU, contiguous env ranges, ONE gather of (obs, reward, done) per step.

The reference has no distributed layer (one env per process, engine_utils.py:8-15).  Here GPU g of G owns environments
[g*N/G, (g+1)*N/G); maps are replicated (immutable); the only exchange is the per-step gather of one packed fp32 row per
env, [A*D obs | A reward | A done].  The engine writes that row itself (pgd_step_packed) straight into this rank's slice of
the receive buffer, so nothing is packed or copied before the exchange.  Two transports:

  "root"        torch.distributed gather to rank 0 (the learner): what the north star names, "a single RCCL gather of (obs, reward,
                done) per step".  RCCL runs it as grouped point-to-point transfers: every peer sends its 4.5 MB slice over its own
                direct xGMI link to GPU 0, the seven transfers run in parallel, nobody receives what it does not need.  The result
                is valid on rank 0 only.
  "collective"  torch.distributed all_gather_into_tensor, in place (send = own slice of the receive buffer).  Backend "nccl"
                is RCCL over xGMI on the GPU box; "gloo" serves the CPU / one-GPU plumbing tests (not in place there).
  "peer"        direct peer writes (pgd_gather_* in include/pgdrive_hip.h): every rank maps the receive buffers of its
                peers over HIP IPC and one kernel pushes the rank's slice over all xGMI links at once, followed by a
                sequence flag per sender; the consumer waits on the flags.  No ring: xGMI is point to point.

Receive buffers are double-buffered: the exchange of step t overlaps the kernels of step t+1.
"""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous partition; the first (n_total % world) ranks get one extra env."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def scenario_ids_for(lo, hi, n_scen):
    """env e -> scenario e mod n_scen, independent of the world size."""
    return (np.arange(lo, hi) % n_scen).astype(np.int32)


def pack_width(obs_dim, num_agents=1):
    return num_agents * (obs_dim + 2)


def pack(torch, obs, reward, done, out=None):
    """[n, A, D] obs + [n, A] reward + [n, A] done -> one fp32 row [n, A*(D+2)] (what pgd_step_packed writes on the device;
    used by hosts that produce the three arrays separately, e.g. the CPU tests)."""
    n, A, D = obs.shape
    if out is None:
        out = torch.empty((n, A * (D + 2)), dtype=torch.float32, device=obs.device)
    out[:, :A * D] = obs.reshape(n, A * D)
    out[:, A * D:A * D + A] = reward
    out[:, A * D + A:A * D + 2 * A] = done.to(torch.float32)
    return out


def unpack(packed, obs_dim, num_agents=1):
    n = packed.shape[0]
    A, D = num_agents, obs_dim
    obs = packed[:, :A * D].reshape(n, A, D)
    reward = packed[:, A * D:A * D + A]
    done = packed[:, A * D + A:A * D + 2 * A] > 0.5
    return obs, reward, done


class StepGather:
    """The per-step exchange.  `produce(rows)` must write this rank's packed rows [n_local, W] (asynchronously on the current
    stream is fine): Engine.step_packed on the GPU, oracle + pack() in the CPU tests.

        g = StepGather(torch, dist, n_local, D, A, device, transport="collective")
        b = g.step(lambda rows: eng.step_packed(actions, rows))     # enqueue step + exchange, returns the buffer index
        obs, reward, done = g.result(b)                              # [world * n_local, ...] once the exchange has landed
    """
    def __init__(self, torch, dist, n_local, obs_dim, num_agents=1, device="cpu", transport="collective", nbuf=2,
                 engine_lib=None, exchange_when_alone=False, device_seq=False):
        """exchange_when_alone: run the collective even in a world of one rank (the single-GPU box's RCCL check: communicator
        set-up, in-place all_gather / gather launches, stream ordering against k_step); by default one rank exchanges nothing.
        device_seq (transport "peer"): the sequence numbers of the exchange live on the device, so the calls of a step are the
        same every time and capture_cycle() can put whole cycles of steps into one HIP graph."""
        self.device_seq = bool(device_seq)
        self.torch, self.dist = torch, dist
        on = dist is not None and dist.is_initialized()
        self.world = dist.get_world_size() if on else 1
        self.rank = dist.get_rank() if on else 0
        self.n_local, self.D, self.A = n_local, obs_dim, num_agents
        self.W = pack_width(obs_dim, num_agents)
        self.nbuf = nbuf
        self.transport = transport if (self.world > 1 or (on and exchange_when_alone and transport != "peer")) else "local"
        self.backend = dist.get_backend() if on else "none"
        self.k = 0
        self.pending = [None] * nbuf
        lo = self.rank * n_local
        if self.transport == "peer":
            from . import peer
            self.peer = peer.PeerGather(torch, dist, engine_lib, n_local, self.W, nbuf, device)
            self.recv = self.peer.recv
            self.send = [r[lo:lo + n_local] for r in self.recv]
            self.inplace = True
        elif self.transport == "root":
            self.peer = None
            self.inplace = False
            # the full buffer exists on the root only; the root's own rows are produced into a separate send buffer and copied
            # into their slice by the gather itself (send and receive memory never alias)
            n_recv = self.world * n_local if self.rank == 0 else n_local
            self.recv = [torch.empty((n_recv, self.W), dtype=torch.float32, device=device) for _ in range(nbuf)]
            self.send = [torch.empty((n_local, self.W), dtype=torch.float32, device=device) if self.rank == 0 else r for r in self.recv]
            self.parts = [[r[q * n_local:(q + 1) * n_local] for q in range(self.world)] if self.rank == 0 else None for r in self.recv]
        else:
            self.peer = None
            self.recv = [torch.empty((self.world * n_local, self.W), dtype=torch.float32, device=device) for _ in range(nbuf)]
            # RCCL / NCCL run all_gather in place when the input is the rank's own slice of the output
            self.inplace = self.world == 1 or self.backend == "nccl"
            self.send = [r[lo:lo + n_local] if self.inplace else
                         torch.empty((n_local, self.W), dtype=torch.float32, device=device) for r in self.recv]

    def describe(self):
        if self.transport == "peer":
            return "direct peer writes over HIP IPC (1 push kernel + sequence flags per step, all xGMI links at once)"
        if self.transport == "local":
            return "single rank: rows written in place, no exchange"
        if self.transport == "root":
            return "1 %s gather(obs|reward|done) to rank 0 per step (grouped point-to-point: one direct xGMI link per peer)" % (
                "RCCL" if self.backend == "nccl" else self.backend)
        return "1 %s all_gather_into_tensor(obs|reward|done) per step%s" % (
            "RCCL" if self.backend == "nccl" else self.backend, ", in place" if self.inplace else "")

    def step(self, produce):
        b = self.k % self.nbuf
        self.wait(b)  # buffer b is about to be overwritten: its previous exchange must have completed
        if self.transport == "peer" and self.device_seq:
            self.peer.release(b, 0)  # ack of the buffer's previous generation (none yet: only the counter moves on)
        elif self.transport == "peer" and self.k >= self.nbuf:
            self.peer.release(b, self.k + 1 - self.nbuf)  # the readers of the previous generation were enqueued before this
        produce(self.send[b])
        if self.transport == "collective":
            self.pending[b] = self.dist.all_gather_into_tensor(self.recv[b], self.send[b], async_op=True)
        elif self.transport == "root":
            self.pending[b] = self.dist.gather(self.send[b], gather_list=self.parts[b], dst=0, async_op=True)
        elif self.transport == "peer":
            self.peer.push(b, 0 if self.device_seq else self.k + 1)
            self.pending[b] = 0 if self.device_seq else self.k + 1
        self.k += 1
        return b

    def wait(self, b):
        p = self.pending[b]
        if p is None:
            return
        if self.transport == "peer":
            self.peer.wait(b, p)
        else:
            p.wait()  # stream-level wait on CUDA tensors, blocking on CPU tensors
        self.pending[b] = None

    def drain(self):
        for b in range(self.nbuf):
            self.wait(b)

    def capture_cycle(self, produces):
        """One HIP graph over len(produces) consecutive steps (a multiple of nbuf; `produces[i]` writes the rows of step i, e.g.
        `lambda rows: eng.step_packed(actions[i], rows)` with a STATIC action tensor): step kernel, exchange and the waits between
        them are enqueued by one `cycle.replay()` per cycle instead of four or five host calls per step -- at 18 us per step the
        host is otherwise the slowest stage of the pipeline (bench.py `gather_model.host_enqueue_us_per_step`).  Transport "peer"
        with device_seq=True (kernels only), or an RCCL collective (torch captures NCCL / RCCL collectives; every exchange of the
        cycle is complete when the graph ends, so cycles do not overlap each other).  Returns a GraphCycle: its replay() launches the
        graph AND moves the step counter on by one cycle (the buffer parity of later eager steps depends on it)."""
        t = self.torch
        n = len(produces)
        assert n > 0 and n % self.nbuf == 0 and self.k % self.nbuf == 0, "whole buffer rounds only"
        assert self.transport != "peer" or self.device_seq, "transport peer: construct with device_seq=True"
        if self.transport in ("root", "collective") and self.backend not in ("nccl", "none"):
            raise RuntimeError("backend %s runs its collectives on the host: they cannot be captured in a HIP graph" % self.backend)
        self.drain()
        dev = self.recv[0].device
        t.cuda.synchronize(dev)
        g = t.cuda.CUDAGraph()
        with t.cuda.graph(g):
            for pr in produces:
                self.step(pr)
            if self.transport != "peer":
                self.drain()  # the collectives' own streams rejoin the capture
        return GraphCycle(self, g, n)

    def result(self, b):
        """(obs, reward, done) of all ranks' envs; with transport "root" on rank 0 only (the other ranks get their own rows)."""
        self.wait(b)
        return unpack(self.recv[b], self.D, self.A)

    # -- self-check of the exchange (bench.py: `gather_ok`) ------------------------------------------------------------------
    @staticmethod
    def row_checksum(torch, rows):
        """Two exact, order-independent 64-bit integer sums over the BIT PATTERNS of `rows` [n, W] fp32 (plain and weighted by
        position; int64 wrap-around is deterministic): equal rows give equal sums, a flipped bit, a swapped pair of rows or
        a slice that landed at the wrong offset changes them."""
        bits = rows.contiguous().view(torch.int32).to(torch.int64).reshape(-1)
        w = (torch.arange(bits.numel(), dtype=torch.int64, device=bits.device) % 65521) + 1
        return torch.stack([bits.sum(), (bits * w).sum()])

    def validate(self, produce, corrupt=None):
        """One more step + exchange, then every rank's checksum of the rows it PRODUCED is compared with the checksum of the slice
        that ARRIVED for it (on rank 0 with transport "root", on every rank otherwise).  Returns (ok, detail) on every rank
        (the verdict is all-reduced).  `corrupt(recv_buffer)` lets a test damage what arrived before the comparison."""
        t, dist = self.torch, self.dist
        b = self.step(produce)
        self.wait(b)
        dev = self.recv[b].device
        if dev.type == "cuda":
            t.cuda.synchronize(dev)
        mine = self.row_checksum(t, self.send[b])  # (before the test hook: in-place transports send from the receive buffer)
        if corrupt is not None:
            corrupt(self.recv[b])
        if self.transport == "local":
            got = self.row_checksum(t, self.recv[b][:self.n_local])
            ok = bool(t.equal(mine, got))
            return ok, dict(ranks_checked=1, mismatched_ranks=[] if ok else [0])
        sums = [t.zeros(2, dtype=t.int64, device=dev) for _ in range(self.world)]
        dist.all_gather(sums, mine)
        bad = []
        if self.transport != "root" or self.rank == 0:
            for q in range(self.world):
                got = self.row_checksum(t, self.recv[b][q * self.n_local:(q + 1) * self.n_local])
                if not bool(t.equal(got, sums[q])):
                    bad.append(q)
        flag = t.tensor([len(bad)], dtype=t.int64, device=dev)
        dist.all_reduce(flag)
        return int(flag.item()) == 0, dict(ranks_checked=self.world, mismatched_ranks=bad,
                                           checked_on="rank 0" if self.transport == "root" else "every rank")

    @property
    def gather_mem(self):
        """Memory kind of the peer transport's receive block: "fine" (fine-grained, what the protocol wants) or "coarse" (the
        runtime refused it: plain hipMalloc); None for the RCCL transports (RCCL owns its staging buffers)."""
        return self.peer.mem_kind if self.peer is not None else None

    def close(self):
        if self.peer is not None:
            self.peer.close()
            self.peer = None


class GraphCycle:
    """A captured cycle of steps (StepGather.capture_cycle).  replay() = graph launch + the gatherer's step bookkeeping, in one call,
    so that the two cannot drift apart (a forgotten `replayed()` left the buffer parity of later eager steps wrong)."""
    def __init__(self, gatherer, graph, steps):
        self.gatherer, self.graph, self.steps = gatherer, graph, steps

    def replay(self):
        self.graph.replay()
        self.gatherer.k += self.steps
