"""Env-sharded data parallelism: one process per GPU, contiguous env ranges, ONE gather of (obs, reward, done) per step.

The reference has no distributed layer (one env per process, engine_utils.py:8-15).  Here GPU g of G owns environments
[g*N/G, (g+1)*N/G); maps are replicated (immutable); the only exchange is the per-step gather, done with a single
torch.distributed all_gather_into_tensor on a packed fp32 row per env (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).
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
    """[n, A, D] obs + [n, A] reward + [n, A] done -> one fp32 row [n, A*(D+2)] (single collective per step)."""
    n, A, D = obs.shape
    if out is None:
        out = torch.empty((n, A * (D + 2)), dtype=torch.float32, device=obs.device)
    out[:, :A * D] = obs.reshape(n, A * D)
    out[:, A * D:A * D + A] = reward
    out[:, A * D + A:] = done.to(torch.float32)
    return out


def unpack(packed, obs_dim, num_agents=1):
    n = packed.shape[0]
    A, D = num_agents, obs_dim
    obs = packed[:, :A * D].reshape(n, A, D)
    reward = packed[:, A * D:A * D + A]
    done = packed[:, A * D + A:] > 0.5
    return obs, reward, done


class StepGather:
    """Owns the send / receive buffers of the per-step collective (equal shard sizes required by all_gather_into_tensor)."""
    def __init__(self, torch, dist, n_local, obs_dim, num_agents=1, device="cpu"):
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.n_local, self.D, self.A = n_local, obs_dim, num_agents
        w = pack_width(obs_dim, num_agents)
        self.send = torch.empty((n_local, w), dtype=torch.float32, device=device)
        self.recv = torch.empty((self.world * n_local, w), dtype=torch.float32, device=device)

    def __call__(self, obs, reward, done):
        pack(self.torch, obs, reward, done, out=self.send)
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.recv, self.send)
        else:
            self.recv.copy_(self.send)
        return unpack(self.recv, self.D, self.A)
