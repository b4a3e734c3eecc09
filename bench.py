#!/usr/bin/env python
"""bench.py — env-steps/sec of the batched PGDrive step engine (BASELINE.json metric).

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = pgd_step over all local environments: every agent and traffic vehicle advanced 0.1 s + (obs, reward, done)
written.  Workload = BASELINE config C3: 4096 envs/GPU x (1 ego + 16 IDM traffic slots) x 240 lidar beams, PGDrive-v0
maps (seeds 1000..1099, env e -> scenario e mod 100), actions uniform(-1,1) from numpy default_rng(0), pre-generated
on the device, auto-reset on done.  Weak scaling: each rank owns 4096 envs.  Environments are independent, so env.step()
has no exchange step: ranks share nothing in the timed region (a data-parallel learner consumes its own shard).  `--gather`
adds the optional learner-side exchange -- one RCCL all_gather of (obs, reward, done) per step over xGMI, double-buffered
-- inside the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic HBM bytes per env-step for this build's record layout:
#   state r/w   2 * V * (23 f32 + 7 i32) * 4 B  + env ints 2*5*4
#   actions 8*A, spawn params read V*48, obs write 4*A*D, reward/done/flags 9*A,
#   k_observe re-read of 7 floats/vehicle
PROF_STRIDE = 16


def algorithmic_bytes(A, T, D):
    V = A + T
    k_step = 2 * V * (23 + 7) * 4 + 40 + 8 * A + V * 48 + 9 * A
    k_obs = V * (5 * 4 + 8) + 4 * A * D + A * 12 * 4
    fused = k_step + 4 * A * D  # observation fused into k_step: no re-read of the vehicle records, obs row written once
    return k_step, k_obs, fused


def load_traffic(N, args):
    """HBM bytes per k_step launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected in separate
    runs of this same command, profiles/r01_pmc_traffic.json); null when the workload differs from the profiled one."""
    p = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(p):
        return None
    t = json.load(open(p))
    if (t.get("envs"), t.get("traffic"), t.get("lasers")) != (N, args.traffic, args.lasers):
        return None
    return t.get("bytes_per_launch")


def cpu_baseline(descs, args, seconds=10.0):
    """Oracle (scalar C restatement) on a bounded sample of the same workload: 1 thread (the reported baseline) and, in
    `all_cores`, OpenMP over envs on every host core."""
    from oracle import orc
    from pgdrive_amd import _abi, mapdata, scenario
    n = 256
    sel = list(descs[:16])
    mb = mapdata.MapBank(sel)
    sb = scenario.ScenarioBank(sel, [d["seed"] for d in sel], num_agents=1, num_traffic=args.traffic)
    cfg = _abi.make_config(n, num_agents=1, num_traffic=args.traffic, num_lasers=args.lasers)
    o = orc.Oracle(cfg, mb, sb)
    o.reset(np.arange(n) % len(sel))
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(32, n, 1, 2)).astype(np.float32)
    o.step(acts[0])
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        o.step(acts[k % 32])
        k += 1
    dt = time.perf_counter() - t0
    o.close()
    # all host cores: the full 4096-env workload, OpenMP over envs inside one parallel region
    cores = min(len(os.sched_getaffinity(0)), 128)
    allc = None
    if cores > 1:
        n2 = args.envs
        cfg2 = _abi.make_config(n2, num_agents=1, num_traffic=args.traffic, num_lasers=args.lasers)
        o2 = orc.Oracle(cfg2, mb, sb)
        o2.reset(np.arange(n2) % len(sel))
        ring = rng.uniform(-1, 1, size=(8, n2, 1, 2)).astype(np.float32)
        o2.run(ring, 2, cores)
        t1 = time.perf_counter()
        k2 = 0
        while time.perf_counter() - t1 < seconds / 2:
            o2.run(ring, 8, cores)
            k2 += 8
        dt2 = time.perf_counter() - t1
        o2.close()
        allc = dict(value=n2 * k2 / dt2, unit="env-steps/s", cores=cores,
                    sample="%d envs x %d steps, OpenMP static over envs" % (n2, k2))
    return dict(value=n * k / dt, unit="env-steps/s", cores=1, kind="port",
                sample="%d envs x %d steps of the C3 workload (16 maps), oracle/pgd_oracle.c fp64 (bicycle restatement, "
                       "not Bullet), 1 thread" % (n, k),
                all_cores=allc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--traffic", type=int, default=16)
    ap.add_argument("--lasers", type=int, default=240)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true",
                    help="N>1: add the learner-side exchange, one RCCL all_gather of (obs, reward, done) per step")
    ap.add_argument("--no-gather", action="store_true", help="(default behaviour; kept for old command lines)")
    ap.add_argument("--actions", default="uniform", choices=["uniform", "straight"],
                    help="uniform(-1,1) (the metric's stream) or drive straight [0,1] with small steering noise (SURVEY 8d)")
    ap.add_argument("--workload", default="c3", choices=["c3", "c5"],
                    help="c3: single-agent PGDrive-v0 (the metric); c5: multi-agent roundabout, --agents agents per env")
    ap.add_argument("--agents", type=int, default=8)
    ap.add_argument("--engines", type=int, default=1,
                    help="E independent engines of --envs environments each, on their own streams, stepped round-robin "
                         "(asynchronous vector-env groups): consecutive steps of different engines overlap on the GPU")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pgdrive_amd import _abi, bank, mapdata, scenario
    from pgdrive_amd.engine import Engine

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local_rank = local_rank % torch.cuda.device_count()  # plumbing tests (gloo) may oversubscribe one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        kw = dict(device_id=dev) if args.backend == "nccl" else {}  # binds the RCCL communicator to this rank's GPU
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world, **kw)

    N = args.envs
    if args.workload == "c5":  # BASELINE config 5: multi-agent roundabout (reported next to the metric, never as the metric)
        from pgdrive_amd import mapgen
        A = args.agents
        args.traffic, args.lasers = 0, 72
        descs = [mapgen.generate_ma_roundabout()]
        mb = mapdata.MapBank(descs)
        sb = scenario.MarlScenarioBank(descs[0], num_agents=A, n_variants=16, seed=rank)
        cfg = _abi.make_config(N, num_agents=A, num_traffic=0, num_lasers=72, num_others=0, lidar_dist=40.0, multi_agent=True,
                               horizon=1000, agent_limit=A, respawn_places=sb.P, respawn_dests=sb.Dn,
                               out_of_road_penalty=10.0, crash_vehicle_penalty=10.0, crash_object_penalty=10.0,
                               delay_done=25, auto_reset=1, resample_scenario=1, seed=1234 + rank)
        n_scen = len(sb.scenarios)
    else:
        A = 1
        descs = bank.get_descriptions(range(1000, 1100))  # generated on the host by our own BIG (pgdrive_amd/mapgen.py)
        mb = mapdata.MapBank(descs)
        sb = scenario.ScenarioBank(descs, [d["seed"] for d in descs], num_agents=1, num_traffic=args.traffic)
        cfg = _abi.make_config(N, num_agents=A, num_traffic=args.traffic, num_lasers=args.lasers, auto_reset=1,
                               seed=1234 + rank)
        n_scen = len(descs)
    eng = Engine(cfg, mb, sb, device=local_rank)
    D = eng.D
    eng.reset((np.arange(N) + rank * N) % n_scen)
    extra = []  # --engines E: E - 1 more engines with their own streams, seeds and scenario offsets
    for j in range(1, max(1, args.engines)):
        import copy
        cfg_j = copy.copy(cfg)
        cfg_j.seed = cfg.seed + 1000 * j
        ej = Engine(cfg_j, mb, sb, device=local_rank)
        ej.reset((np.arange(N) + (rank * args.engines + j) * N) % n_scen)
        extra.append(ej)

    rng = np.random.default_rng(rank)  # rank 0 == default_rng(0)
    CYC = 64
    if args.actions == "uniform":
        acts = rng.uniform(-1, 1, size=(CYC, N, A, 2)).astype(np.float32)
    else:  # "drive straight" (profile_pgdrive.py:16): full throttle, a little steering noise so that episodes differ
        acts = np.zeros((CYC, N, A, 2), dtype=np.float32)
        acts[..., 0] = rng.normal(0, 0.05, size=(CYC, N, A))
        acts[..., 1] = 1.0
    actions = torch.from_numpy(acts).to(dev)

    gather = world > 1 and args.gather and not args.no_gather
    if gather:
        # one exchange per step: obs | reward | done packed into a single fp32 row per env, double-buffered:
        # the all_gather of step t (RCCL's own stream) overlaps the kernels of step t+1; buffer b is re-used at step t+2
        # only after its gather has completed
        bufs = [eng.make_outputs() for _ in range(2)]
        packs = [torch.empty((N, A * (D + 2)), dtype=torch.float32, device=dev) for _ in range(2)]
        gathered = [torch.empty((world * N, A * (D + 2)), dtype=torch.float32, device=dev) for _ in range(2)]
        pending = [None, None]

    def one_step(k):
        if not gather:
            eng.step(actions[k % CYC])
            for j, ej in enumerate(extra):  # each engine enqueues on its own stream: no ordering between engines
                with torch.cuda.stream(ej.stream):
                    ej.step(actions[(k + 7 * (j + 1)) % CYC])
            return
        b = k % 2
        if pending[b] is not None:
            pending[b].wait()  # stream-level wait: step k may overwrite what the gather of step k-2 was reading
        obs, rew, done, flags = eng.step(actions[k % CYC], out=bufs[b])
        pack = packs[b]
        pack[:, :A * D] = obs.view(N, A * D)
        pack[:, A * D:A * D + A] = rew
        pack[:, A * D + A:] = done.to(torch.float32)
        pending[b] = dist.all_gather_into_tensor(gathered[b], pack, async_op=True)

    def drain():
        if gather:
            for w in pending:
                if w is not None:
                    w.wait()

    with torch.cuda.stream(eng.stream):
        for k in range(args.warmup):
            one_step(k)
        drain()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        # HIP events bracket groups of PROF_STRIDE consecutive k_step launches over the whole timed region; the average
        # launch duration is group time / PROF_STRIDE (two event packets around EVERY launch leave the command processor
        # idle between back-to-back kernels and slow the thing being measured: 102 -> 125 M env-steps/s without them)
        eng.profile_begin(args.steps // PROF_STRIDE + 1, stride=PROF_STRIDE)
        t0 = time.perf_counter()
        for k in range(args.steps):
            one_step(args.warmup + k)
        drain()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        prof = eng.profile_end()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_env_steps = float(N) * world * args.steps * max(1, args.engines)
        value = total_env_steps / elapsed
        b_step, b_obs, b_fused = algorithmic_bytes(A, args.traffic, D)
        fused = prof["k_observe_ms"] == 0.0  # pgd_step ran the observation inside k_step (one env per wave)
        if fused:
            b_step, b_obs = b_fused, 0
        dom = "k_observe" if prof["k_observe_ms"] >= prof["k_step_ms"] else "k_step"
        dom_ms = max(prof["k_observe_ms"], prof["k_step_ms"])
        dom_bytes = (b_obs if dom == "k_observe" else b_step) * N
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        out = {
            "metric": "env-steps/sec (whole node) at 4096 envs x 240 lidar beams",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": ("C3: %d envs/GPU x (1 ego + %d IDM traffic slots) x %d lidar beams, PGDrive-v0 maps "
                             "seeds 1000-1099, %s actions, auto-reset" % (
                                 N, args.traffic, args.lasers, "uniform(-1,1)" if args.actions == "uniform" else "drive-straight"))
                if args.workload == "c3" else
                ("C5: %d envs/GPU x %d agents, multi-agent roundabout, 72 beams x 40 m, %s actions, respawn, auto-reset; "
                 "agent-steps/s = value x %d" % (N, A, args.actions, A)),
                **({"note": "%d independent engines x %d envs on their own streams, stepped round-robin: consecutive steps "
                            "of different engines overlap (asynchronous vector-env groups)" % (args.engines, N)}
                   if args.engines > 1 else {}),
                "envs_per_gpu": N * max(1, args.engines), "global_envs": N * world * max(1, args.engines), "obs_dim": D,
                "engines_per_gpu": max(1, args.engines),
                "parallelism": "env-sharded dp%d%s" % (world, " + 1 RCCL all_gather(obs,reward,done)/step, double-buffered" if gather else ", no data-path collective"),
            },
            "roofline": {
                "bound": "hbm", "kernel": dom + (" (observation fused)" if fused else ""), "achieved": achieved,
                "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": load_traffic(N, args),
                "bytes_per_env_step": {"k_step": b_step, "k_observe": b_obs},
                "k_step_ms": prof["k_step_ms"], "k_observe_ms": prof["k_observe_ms"], "events": prof["count"],
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(descs, args) if args.workload == "c3" else None
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
