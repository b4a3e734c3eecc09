#!/usr/bin/env python
"""bench.py — env-steps/sec of the batched PGDrive step engine (BASELINE.json metric).

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python bench.py --gpus 8 ...                        (spawns the 8 ranks itself when WORLD_SIZE is not set)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = pgd_step over all local environments: every agent and traffic vehicle advanced 0.1 s + (obs, reward, done)
written.  Workload = BASELINE config C3: 4096 envs/GPU x (1 ego + 16 IDM traffic slots) x 240 lidar beams, PGDrive-v0
maps (seeds 1000..1099, env e -> scenario e mod 100), actions uniform(-1,1) from numpy default_rng(0), pre-generated
on the device, auto-reset on done.  Weak scaling: each rank owns 4096 envs.

Steady state.  The first ~1000 steps after a reset are cheaper than the rest (the trigger traffic is still parked), so a
short run would report an early-episode number.  The bench therefore ALWAYS pre-rolls at least PREROLL_MIN steps before
the timed region and times at least TIMED_MIN steps, whatever --warmup / --steps say; the JSON line carries the requested
counts ("steps", "warmup") and the counts actually run ("steps_timed", "warmup_run"); `ms_per_step` and `value` refer to
the timed steps.  --exact turns the floors off.

Rows.  At N = 1 the same invocation also times the LOADED workloads next to the metric's (whose uniform(-1,1) stream leaves the
ego crawling and most traffic parked): the scripted lane-keeping ego, respawn-mode traffic (every IDM vehicle drives), BASELINE
config 5 (multi-agent roundabout, 8 agents, 240 and 72 beams; 40 agents = the reference's default), and 32768 envs on the one GPU
(config 4's per-node size) -- printed as `rows: [...]`, each with its own roofline (shorter windows: 1500 + 2048 steps;
--no-rows skips them).  `roofline.frac` is charged for the bytes that MOVE: the HBM bytes the rocprofv3 counters saw for that
row (profiles/r*_pmc_*.json, matched by workload) or, without a counter file, the record bytes of the vehicles that drove
(`frac_active`); the nominal formula that charges all V records read + written stays next to it as `frac_nominal`.

N > 1.  Environments are independent: the step itself has no exchange.  The north star adds ONE gather of
(obs, reward, done) per step over xGMI; both are measured in the same invocation: `value` is WITH the per-step gather
(pgd_step_packed writes the packed row straight into the send buffer, one RCCL gather to rank 0 per step -- the learner's GPU;
RCCL runs it as seven point-to-point transfers over seven different xGMI links -- double-buffered), `value_replicas` is without it
(a data-parallel learner that consumes its own shard).  --transport collective gathers to EVERY rank instead
(all_gather_into_tensor), --transport peer uses direct peer writes over HIP IPC (pgdrive_amd/peer.py).
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HOST_THREADS = len(os.sched_getaffinity(0))  # read before any OpenMP runtime binds the main thread to one core (OMP_PROC_BIND)

PROF_STRIDE = 64      # k_step launches per HIP-event group (an event pair costs a launch gap: 16 per group took 0.4 us off every step)
PREROLL_MIN = 1500    # steps before the timed region (steady-state traffic)
TIMED_MIN = 4096      # timed steps (>= 64 event groups of PROF_STRIDE launches)


# Algorithmic HBM bytes per env-step for this build's record layout:
#   vehicle records  V x 128 B read + V x 128 B written (one cache line per slot: ABI fields + carried derived state)
#   env row 32 B read + written, per-env map header 64 B read, actions 8*A, spawn parameters V*48 (pose .. max_speed),
#   reward / done / flags 9*A, observation row 4*A*D written (fused) -- k_observe (stand-alone): V records' 28 B + row + view
def algorithmic_bytes(A, T, D):
    V = A + T
    k_step = 2 * V * 128 + 64 + 64 + 8 * A + V * 48 + 9 * A
    k_obs = V * (5 * 4 + 8) + 4 * A * D + A * 12 * 4
    fused = k_step + 4 * A * D  # observation fused into k_step: no re-read of the vehicle records, obs row written once
    return k_step, k_obs, fused


def load_traffic(N, args):
    """HBM bytes per k_step launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected in separate
    runs of this same command, newest profiles/r*_pmc_traffic.json whose workload matches); null when none matches."""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            t = json.load(open(p))
        except Exception:
            continue
        if (t.get("envs"), t.get("traffic"), t.get("lasers"), t.get("actions", "uniform"), t.get("traffic_mode", "trigger"),
                t.get("workload", "c3"), t.get("agents", 1)) == \
                (N, args.traffic, args.lasers, args.actions, args.traffic_mode, args.workload, args.agents if args.workload == "c5" else 1):
            return t.get("bytes_per_launch"), os.path.basename(p), t.get("bytes_per_launch_k_observe")
    return None, None, None


def cpu_baseline(descs, args, seconds=6.0):
    """Oracle (scalar C restatement) on a bounded sample of the same workload: 1 thread (the reported baseline: 256 envs, three
    windows, median) and `threads_curve`: the full 4096-env workload with OpenMP over envs on 1 / 8 / 32 / all host threads
    (pinned: OMP_PROC_BIND / OMP_PLACES are set in main() before libgomp loads; one window of >= 5 s each, best of two for
    the all-threads point).  `all_cores` is the last point of that curve; when it is less than a quarter of what its thread
    count would give at the 8-thread efficiency, the line says so ("host oversubscribed") instead of presenting it as a
    baseline."""
    from oracle import orc
    from pgdrive_amd import _abi, mapdata, scenario
    n = 256
    sel = list(descs[:16])
    mb = mapdata.MapBank(sel)
    sb = scenario.ScenarioBank(sel, [d["seed"] for d in sel], num_agents=1, num_traffic=args.traffic,
                               traffic_mode=args.traffic_mode)
    cfg = _abi.make_config(n, num_agents=1, num_traffic=args.traffic, num_lasers=args.lasers)
    o = orc.Oracle(cfg, mb, sb)
    o.reset(np.arange(n) % len(sel))
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(32, n, 1, 2)).astype(np.float32)
    o.step(acts[0])
    # three timed windows, the median is reported (host noise of +-25 % between driver runs was seen with one window)
    rates, k = [], 0
    for w in range(3):
        t0 = time.perf_counter()
        k0 = k
        while time.perf_counter() - t0 < seconds / 3:
            o.step(acts[k % 32])
            k += 1
        rates.append(n * (k - k0) / (time.perf_counter() - t0))
    rate1 = float(np.median(rates))
    o.close()
    avail = HOST_THREADS
    quota = None
    try:  # cgroup v2 CPU quota ("max 100000" = none)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    # what the process may use: the affinity mask, capped by the cgroup's CPU quota (the GPU box shows 256 threads and grants 16
    # CPUs' worth of time: more runnable threads than that only take turns)
    limit = min(avail, 128)
    if quota:
        limit = max(1, min(limit, int(round(quota))))
    allc, curve = None, []
    if limit > 1:
        n2 = args.envs
        cfg2 = _abi.make_config(n2, num_agents=1, num_traffic=args.traffic, num_lasers=args.lasers)
        o2 = orc.Oracle(cfg2, mb, sb)
        o2.reset(np.arange(n2) % len(sel))
        ring = rng.uniform(-1, 1, size=(8, n2, 1, 2)).astype(np.float32)
        o2.run(ring, 2, limit)
        points = sorted(set(t for t in (1, 8, 32, limit, 2 * limit) if t <= min(avail, 128) and (t <= 2 * limit)))
        for th in points:
            best = 0.0
            for rep in range(2 if th == limit else 1):
                t1 = time.perf_counter()
                k2 = 0
                while time.perf_counter() - t1 < (5.0 if th > 1 else 3.0):
                    o2.run(ring, 4 if th < 8 else 16, th)
                    k2 += 4 if th < 8 else 16
                best = max(best, n2 * k2 / (time.perf_counter() - t1))
            curve.append(dict(threads=th, value=round(best), speedup_vs_1=None))
        o2.close()
        for c in curve:
            c["speedup_vs_1"] = round(c["value"] / curve[0]["value"], 2)
        at = next(c for c in curve if c["threads"] == limit)
        # "oversubscribed": the granted CPUs do not deliver -- less than half of a linear speed-up at the limit
        oversub = at["speedup_vs_1"] < 0.5 * limit
        allc = dict(value=float(at["value"]), unit="env-steps/s", cores=limit, host_threads_visible=avail,
                    cgroup_cpu_quota=quota, threads_curve=curve, host_oversubscribed=bool(oversub),
                    sample="%d envs, OpenMP dynamic over envs, threads pinned (OMP_PROC_BIND=close, OMP_PLACES=cores), one "
                           ">= 5 s window per point (best of two at the limit); cores = min(affinity mask, cgroup CPU quota), "
                           "the point beyond it shows that more threads only take turns" % n2,
                    **({"note": "host oversubscribed: %d threads give %.1fx of one thread -- the granted CPUs are shared with "
                                "other tenants; not a usable all-cores baseline" % (limit, at["speedup_vs_1"])} if oversub else {}))
    return dict(value=rate1, unit="env-steps/s", cores=1, kind="port", windows=[round(r) for r in rates],
                sample="%d envs x %d steps of the C3 workload (16 maps) in 3 windows (median), oracle/pgd_oracle.c fp64 (bicycle "
                       "restatement, not Bullet), 1 thread" % (n, k),
                all_cores=allc)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--exact", action="store_true",
                    help="run exactly --warmup / --steps (no steady-state floors): for plumbing tests and quick A/B runs")
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--traffic", type=int, default=16)
    ap.add_argument("--maps", type=int, default=100,
                    help="c3: number of PGDrive-v0 maps (seeds 1000 ...) the envs are spread over; the metric's workload is 100")
    ap.add_argument("--lasers", type=int, default=None,
                    help="lidar beams: default 240 for c3; for c5 72 x 40 m (the reference's multi-agent default) unless given "
                         "(BASELINE.md C5 is --lasers 240)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rows", action="store_true",
                    help="N = 1: only the metric's workload, without the loaded rows (expert / respawn / C5 / 32768 envs)")
    ap.add_argument("--rows", default=None,
                    help="comma-separated subset of the rows to run (names as printed in `rows`), default all")
    ap.add_argument("--mode", default="both", choices=["both", "gather", "replicas"],
                    help="N>1: time the step with the per-step gather (value), without it (value_replicas), or both")
    ap.add_argument("--gather", action="store_true", help="(old flag) same as --mode gather")
    ap.add_argument("--no-gather", action="store_true", help="(old flag) same as --mode replicas")
    ap.add_argument("--transport", default="root", choices=["root", "collective", "peer"],
                    help="the per-step gather: RCCL gather to rank 0 (default), RCCL all_gather_into_tensor, or direct peer writes over HIP IPC")
    ap.add_argument("--actions", default="uniform", choices=["uniform", "straight", "expert"],
                    help="uniform(-1,1) (the metric's stream), drive straight [0,1] with small steering noise (SURVEY 8d), or "
                         "expert: the scripted lane-keeping policy of the library (pgd_lane_keep_actions, 30 km/h cruise) on the "
                         "last observation -- the ego keeps driving, traffic gets triggered, episodes end by arrival")
    ap.add_argument("--traffic-mode", default="trigger", choices=["trigger", "respawn", "hybrid"],
                    help="TrafficManager mode (traffic_manager.py:19-27): trigger (the reference default and the metric), respawn "
                         "(every traffic vehicle drives from the first step of an episode), hybrid")
    ap.add_argument("--workload", default="c3", choices=["c3", "c5"],
                    help="c3: single-agent PGDrive-v0 (the metric); c5: multi-agent roundabout, --agents agents per env")
    ap.add_argument("--agents", type=int, default=8)
    ap.add_argument("--engines", type=int, default=1,
                    help="E independent engines of --envs environments each, on their own streams, stepped round-robin "
                         "(asynchronous vector-env groups): consecutive steps of different engines overlap on the GPU")
    ap.add_argument("--groups", type=int, default=1,
                    help="split the --envs environments of the engine into G asynchronous env groups (pgd_set_groups / "
                         "pgd_step_group): one 'step' = every group stepped once, the groups' launches overlap")
    ap.add_argument("--step-n", type=int, default=1,
                    help="open-loop variant: K steps of the action ring per pgd_step_n call (observation only for the last state of "
                         "each call); reported next to the metric, never as the metric (the metric is the closed loop)")
    ap.add_argument("--topdown", action="store_true",
                    help="c3 with the top-down image observation (TopDownPGDriveEnv: 84 x 84 x 5, lidar off): pgd_step + "
                         "pgd_observe_topdown per step; reported next to the metric, never as the metric")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="plumbing tests: allow more ranks than GPUs (ranks share devices; needs --backend gloo)")
    ap.add_argument("--graph", action="store_true",
                    help="N > 1, gather pass: capture step + exchange of a whole action cycle (64 steps) in ONE HIP graph per rank and "
                         "replay it (transport peer: device-side sequence numbers; RCCL transports: torch captures the collectives) -- "
                         "the host then costs one launch per 64 steps instead of several calls per step")
    ap.add_argument("--corrupt-gather", action="store_true",
                    help="tests only: flip one value of the received rows before the gather's self-check (gather_ok must read false)")
    args = ap.parse_args(argv)
    if args.gather:
        args.mode = "gather"
    if args.no_gather:
        args.mode = "replicas"
    if args.lasers is None:
        args.lasers = (0 if args.topdown else 240) if args.workload == "c3" else 72
    return args


# The loaded rows timed next to the metric's workload at N = 1 (name, overrides of the command line).  Windows: 1500 + 2048 steps
# (C5: 1000 + 1536: the roundabout fills up over the first thousand steps).
ROWS = [
    ("c3_expert", dict(actions="expert")),
    ("c3_respawn", dict(traffic_mode="respawn")),
    ("c3_expert_respawn", dict(actions="expert", traffic_mode="respawn")),
    ("c5_8x240", dict(workload="c5", agents=8, lasers=240, warmup=1000, steps=1536)),
    ("c5_8x72", dict(workload="c5", agents=8, lasers=72, warmup=1000, steps=1536)),
    ("c5_40x72", dict(workload="c5", agents=40, lasers=72, warmup=1000, steps=2048)),
    ("c3_32768", dict(envs=32768, warmup=1500, steps=512)),
]

XGMI_LINK_GBPS = 153.0  # per direction and link (MI355X_MICROARCH.md); 7 links per GPU, point to point


def measure(args, rank, world, local_rank, with_cpu_baseline=True):
    """One workload: build the engine, pre-roll, time, read the work statistics back; returns the JSON line as a dict on rank 0
    (None elsewhere).  The process group, if any, already exists."""
    import torch
    import torch.distributed as dist
    from pgdrive_amd import _abi, bank, mapdata, scenario
    from pgdrive_amd import dist as pdist
    from pgdrive_amd.engine import Engine

    n_dev = torch.cuda.device_count()
    local_rank = local_rank % n_dev
    dev = torch.device("cuda", local_rank)

    N = args.envs
    if args.workload == "c5":  # BASELINE config 5: multi-agent roundabout (reported next to the metric, never as the metric)
        from pgdrive_amd import mapgen
        A = args.agents
        args.traffic = 0
        descs = [mapgen.generate_ma_roundabout()]
        mb = mapdata.MapBank(descs)
        sb = scenario.MarlScenarioBank(descs[0], num_agents=A, n_variants=16, seed=rank)
        cfg = _abi.make_config(N, num_agents=A, num_traffic=0, num_lasers=args.lasers, num_others=0, lidar_dist=40.0, multi_agent=True,
                               horizon=1000, agent_limit=A, respawn_places=sb.P, respawn_dests=sb.Dn,
                               out_of_road_penalty=10.0, crash_vehicle_penalty=10.0, crash_object_penalty=10.0,
                               delay_done=25, auto_reset=1, resample_scenario=1, seed=1234 + rank, env_base=rank * N)
        n_scen = len(sb.scenarios)
    else:
        A = 1
        descs = bank.get_descriptions(range(1000, 1000 + args.maps))  # generated on the host by our own BIG (pgdrive_amd/mapgen.py)
        mb = mapdata.MapBank(descs)
        sb = scenario.ScenarioBank(descs, [d["seed"] for d in descs], num_agents=1, num_traffic=args.traffic,
                                   traffic_mode=args.traffic_mode)
        cfg = _abi.make_config(N, num_agents=A, num_traffic=args.traffic, num_lasers=args.lasers, auto_reset=1,
                               seed=1234 + rank, env_base=rank * N)
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
    CYC = 64 if N <= 8192 else 8
    if args.actions in ("uniform", "expert"):  # (expert: the ring only feeds the pre-roll of engines without an observation yet)
        acts = rng.uniform(-1, 1, size=(CYC, N, A, 2)).astype(np.float32)
    else:  # "drive straight" (profile_pgdrive.py:16): full throttle, a little steering noise so that episodes differ
        acts = np.zeros((CYC, N, A, 2), dtype=np.float32)
        acts[..., 0] = rng.normal(0, 0.05, size=(CYC, N, A))
        acts[..., 1] = 1.0
    actions = torch.from_numpy(acts).to(dev)

    warm = args.warmup if args.exact else max(args.warmup, PREROLL_MIN)
    timed = args.steps if args.exact else max(args.steps, TIMED_MIN)
    # N > 1: the replicas pass first (no collective inside its timed loop), then the pass with the per-step gather -- if the
    # gather transport fails on hardware it has never run on, the line still carries the replicas number and says why
    modes = ["replicas"] if world == 1 else (["replicas", "gather"] if args.mode == "both" else [args.mode])
    gatherer = None
    if "gather" in modes:
        gatherer = pdist.StepGather(torch, dist, N, D, A, device=dev, transport=args.transport, engine_lib=eng.L,
                                    device_seq=bool(args.graph and args.transport == "peer"))

    if args.groups > 1:
        eng.set_groups(args.groups)
    if args.topdown:
        eng.enable_topdown()

    expert = args.actions == "expert"
    if expert and (A != 1 or args.groups > 1 or args.step_n > 1 or args.engines > 1 or args.topdown or "gather" in modes):
        raise SystemExit("--actions expert: single-agent closed loop on the rank's own shard only (N > 1: --mode replicas)")
    act_buf = torch.zeros((N, A, 2), dtype=torch.float32, device=dev)

    def step_replica(k):
        if expert:  # closed loop: observation of step k - 1 -> scripted policy (one small launch) -> step k
            eng.lane_keep_actions(act_buf, k)
            eng.step(act_buf)
            return
        if args.groups > 1:
            for g in range(args.groups):  # each group on its own internal stream: the launches overlap
                eng.step_group(g, actions[(k + 5 * g) % CYC])
            return
        if args.step_n > 1:  # one call per K steps; the other K - 1 "steps" of the loop are part of that call
            if k % args.step_n == 0:
                eng.step_n(actions, k % CYC, args.step_n)
            return
        eng.step(actions[k % CYC], want_obs=not args.topdown)
        if args.topdown:
            eng.observe_topdown()
        for j, ej in enumerate(extra):  # each engine enqueues on its own stream: no ordering between engines
            with torch.cuda.stream(ej.stream):
                ej.step(actions[(k + 7 * (j + 1)) % CYC])

    def step_gather(k):
        a = actions[k % CYC]
        gatherer.step(lambda rows: eng.step_packed(a, rows))

    def fence():
        for g in range(args.groups if args.groups > 1 else 0):
            eng.group_sync(g)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    results = {}
    counter = 0
    gather_check = None
    with torch.cuda.stream(eng.stream):
        for k in range(warm):  # pre-roll to steady-state traffic, once, shared by both modes
            step_replica(counter)
            counter += 1
        gather_error = None
        for mode in modes:
            one_step = step_gather if mode == "gather" else step_replica
            try:
                for k in range(16 if not args.exact else min(16, args.warmup)):  # the mode's own buffers / communicator warm-up
                    one_step(counter)
                    counter += 1
                if gatherer is not None and mode == "gather":
                    gatherer.drain()
                fence()
            except Exception as ex:  # noqa: BLE001  (a transport that does not come up: report it, keep the other pass)
                if mode != "gather" or "replicas" not in results:
                    raise
                gather_error = "%s: %s" % (type(ex).__name__, str(ex)[:300])
                print("bench.py: the per-step gather failed on rank %d: %s" % (rank, gather_error), file=sys.stderr, flush=True)
                break
            # HIP events bracket groups of PROF_STRIDE consecutive k_step launches over the whole timed region; the average
            # launch duration is group time / PROF_STRIDE (two event packets around EVERY launch leave the command processor
            # idle between back-to-back kernels and slow the thing being measured)
            profiled = mode == "replicas" and args.step_n == 1  # (the open-loop variant mixes launches with and without the observation)
            # short --exact runs (tests, sweeps of a few hundred steps): smaller groups, so that several of them complete
            stride = PROF_STRIDE if timed >= 8 * PROF_STRIDE else (16 if timed >= 16 else 1)
            if profiled:
                eng.profile_begin(timed // stride + 1, stride=stride)
            graph, cyc = None, 0
            if mode == "gather" and args.graph:
                import math
                cyc = math.gcd(timed, CYC)
                if cyc % gatherer.nbuf == 0 and cyc >= gatherer.nbuf:
                    graph = gatherer.capture_cycle([(lambda rows, a=actions[i]: eng.step_packed(a, rows)) for i in range(cyc)])
                    fence()
            t0 = time.perf_counter()
            if graph is not None:  # one host call per cycle of `cyc` steps
                for c in range(timed // cyc):
                    graph.replay()
                    gatherer.replayed()
                counter += timed
            else:
                for k in range(timed):
                    one_step(counter)
                    counter += 1
            t_enq = time.perf_counter()  # every launch of the timed region has been enqueued: the host's share of the loop
            if mode == "gather":
                gatherer.drain()
            fence()
            t1 = time.perf_counter()
            prof = eng.profile_end() if profiled else None
            elapsed = t1 - t0
            if world > 1:
                t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
            results[mode] = dict(elapsed=elapsed, prof=prof, host_enqueue_s=t_enq - t0, graph_cycle=cyc if graph is not None else 0)
            if mode == "gather":
                # self-check of the exchange, after the timed loop: every rank's checksum of the rows it produced against the
                # checksum of what arrived for it (a transport that delivers wrong rows would otherwise still print a number)
                a_chk = actions[counter % CYC]
                corrupt = None
                if args.corrupt_gather:
                    def corrupt(buf):
                        if buf.shape[0] > N or world == 1:  # the rank that holds the gathered rows
                            buf[buf.shape[0] - 1, 3] += 1.0
                ok, detail = gatherer.validate(lambda rows: eng.step_packed(a_chk, rows), corrupt=corrupt)
                counter += 1
                gather_check = dict(ok=ok, **detail)

    step_kernel = eng.describe_step()  # which k_step instantiation the timed launches were (pgd_describe_step)
    # how much work a step does at this point of the run: 5 snapshots of the state, 50 untimed steps apart
    work = None
    if args.workload == "c3" and N <= 65536:
        from pgdrive_amd import _abi as abi
        act_n, with_t, ep = [], [], []
        with torch.cuda.stream(eng.stream):
            for snap in range(5):
                for k in range(50 if snap else 0):
                    step_replica(counter)
                    counter += 1
                fence()
                f_, i_, ei_ = eng.get_state()
                drv = (i_[abi.SI["STATUS"]][:, A:] == abi.ST_ACTIVE)
                act_n.append(float(drv.sum(axis=1).mean()))
                with_t.append(float(drv.any(axis=1).mean()))
                ep.append(float(ei_[abi.EI["EP_STEPS"]].mean()))
        spd = float(np.abs(f_[abi.SF["SPEED"]][:, 0]).mean() * 3.6)
        work = dict(driving_traffic_mean=float(np.mean(act_n)), envs_with_traffic_frac=float(np.mean(with_t)),
                    episode_step_mean=float(np.mean(ep)), ego_speed_kmh_mean=spd)
    elif args.workload == "c5":  # (the agent population swings with the 1000-step agent horizon: five snapshots, 100 steps apart)
        from pgdrive_amd import _abi as abi
        act_a, pres_a = [], []
        with torch.cuda.stream(eng.stream):
            for snap in range(5):
                for k in range(100 if snap else 0):
                    step_replica(counter)
                    counter += 1
                fence()
                f_, i_, ei_ = eng.get_state()
                st_ = i_[abi.SI["STATUS"]][:, :A]
                act_a.append(float((st_ == abi.ST_ACTIVE).sum(axis=1).mean()))
                pres_a.append(float(((st_ == abi.ST_ACTIVE) | (st_ == abi.ST_DYING)).sum(axis=1).mean()))
        work = dict(active_agents_mean=float(np.mean(act_a)), present_agents_mean=float(np.mean(pres_a)),
                    active_agents_snapshots=[round(v, 2) for v in act_a])

    ranks_ran = world
    rccl_ranks = None
    if world > 1:
        if args.backend == "nccl":  # proves that RCCL itself saw every rank (the first multi-GPU run is a first run)
            assert dist.get_backend() == "nccl"
            rccl_ranks = dist.get_world_size()
        t = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        ranks_ran = int(round(t.item()))

    out = None
    if rank == 0:
        per_step_units = float(N) * world * max(1, args.engines)
        head = "gather" if "gather" in results else "replicas"
        elapsed = results[head]["elapsed"]
        value = per_step_units * timed / elapsed
        out = {
            "metric": "env-steps/sec (whole node) at 4096 envs x 240 lidar beams",
            "value": value, "unit": "env-steps/s", "n_gpus": ranks_ran, "steps": args.steps, "warmup": args.warmup,
            "steps_timed": timed, "warmup_run": warm, "steps_effective": timed, "rccl_ranks": rccl_ranks,
            "ms_per_step": elapsed / timed * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        }
        if world > 1:
            out["value_mode"] = head
            if gather_error:
                out["gather_error"] = gather_error
            for m in results:
                out["value_" + m] = per_step_units * timed / results[m]["elapsed"]
                out["ms_per_step_" + m] = results[m]["elapsed"] / timed * 1e3
            if gather_check is not None:
                out["gather_ok"] = gather_check["ok"]
                out["gather_check"] = gather_check
            if "gather" in results:
                # what the exchange can cost by construction: each peer's slice over its own xGMI link into
                # the root (transport root / peer: the links work in parallel, the slowest is one slice), the all-gather as a ring
                # (per-link bound: (n - 1) slices through every link), and the host's enqueue time per step of this very loop
                slice_bytes = N * pdist.pack_width(D, A) * 4
                link_us = slice_bytes / (XGMI_LINK_GBPS * 1e9) * 1e6
                ring_us = (world - 1) * slice_bytes / (XGMI_LINK_GBPS * 1e9) * 1e6
                k_us = (results.get("replicas", {}).get("prof") or {}).get("k_step_ms", 0.0) * 1e3 or None
                host_us = results["gather"]["host_enqueue_s"] / timed * 1e6
                bound_us = ring_us if args.transport == "collective" else link_us
                cands = [x for x in (k_us, bound_us, host_us) if x]
                out["gather_model"] = {
                    "slice_bytes_per_rank_per_step": slice_bytes, "xgmi_link_GBps": XGMI_LINK_GBPS,
                    "link_bound_us": link_us, "ring_allgather_bound_us": ring_us, "k_step_us": k_us,
                    "host_enqueue_us_per_step": host_us,
                    "host_enqueue_us_per_step_replicas": (results["replicas"]["host_enqueue_s"] / timed * 1e6) if "replicas" in results else None,
                    "hip_graph_steps_per_replay": results["gather"]["graph_cycle"] or None,
                    "predicted_floor_us_per_step": max(cands) if cands else None,
                    "predicted_ceiling_env_steps_per_s": per_step_units / (max(cands) * 1e-6) if cands else None,
                    "note": "the exchange of step t overlaps the kernels of step t + 1 (double-buffered): the step rate is bounded by the "
                            "slowest of kernel, link and host enqueue, not by their sum",
                }
        if head == "gather":
            par = "env-sharded dp%d + %s, double-buffered" % (world, gatherer.describe())
        else:
            par = "env-sharded dp%d, no data-path collective" % world
        out["config"] = {
            "workload": ("C3: %d envs/GPU x (1 ego + %d IDM traffic slots) x %d lidar beams, PGDrive-v0 maps "
                         "seeds 1000-1099, %s actions, auto-reset" % (
                             N, args.traffic, args.lasers, {"uniform": "uniform(-1,1)", "straight": "drive-straight",
                                                            "expert": "scripted lane-keeping (30 km/h)"}[args.actions]) +
                         ("" if args.traffic_mode == "trigger" else ", traffic mode " + args.traffic_mode))
            if args.workload == "c3" else
            ("C5: %d envs/GPU x %d agents, multi-agent roundabout, %d beams x 40 m, %s actions, respawn, auto-reset; "
             "agent-steps/s = value x %d" % (N, A, args.lasers, args.actions, A)),
            **({"note": "%d independent engines x %d envs on their own streams, stepped round-robin: consecutive steps "
                        "of different engines overlap (asynchronous vector-env groups)" % (args.engines, N)}
               if args.engines > 1 else {}),
            **({"active_vehicles_mean": 1.0 + work["driving_traffic_mean"]} if work and "driving_traffic_mean" in work else {}),
            **(work or {}),
            "step_kernel": step_kernel,
            "envs_per_gpu": N * max(1, args.engines), "global_envs": N * world * max(1, args.engines), "obs_dim": D,
            "engines_per_gpu": max(1, args.engines), "env_groups": args.groups,
            **({"open_loop": "pgd_step_n: %d steps of the action ring per call, one observation per call -- NOT the metric's closed "
                             "loop" % args.step_n} if args.step_n > 1 else {}),
            **({"observation": "top-down image 84 x 84 x 5 float32 (pgd_observe_topdown), %.1f MB written per step" % (
                N * 84 * 84 * 5 * 4 / 1e6)} if args.topdown else {}),
            "parallelism": par, "backend": (args.backend if world > 1 else "none"),
            "steady_state": "pre-roll %d steps, %d timed steps (floors %d / %d%s)" % (
                warm, timed, PREROLL_MIN, TIMED_MIN, ", off: --exact" if args.exact else ""),
        }
        prof = results.get("replicas", {}).get("prof")
        if prof is not None:
            b_step, b_obs, b_fused = algorithmic_bytes(A, args.traffic, D)
            fused = prof["k_observe_ms"] == 0.0  # pgd_step ran the observation inside k_step (one env per wave)
            if fused:
                b_step, b_obs = b_fused, 0
            dom = "k_observe" if prof["k_observe_ms"] >= prof["k_step_ms"] else "k_step"
            dom_ms = max(prof["k_observe_ms"], prof["k_step_ms"])
            dom_bytes = (b_obs if dom == "k_observe" else b_step) * N
            nominal = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
            traffic, traffic_src, traffic_obs = load_traffic(N, args)
            if dom == "k_observe":
                traffic = traffic_obs  # the counters' figure of the observation kernel (multi-agent engines with many slots)
            # bytes that MOVE per launch: the counters' figure when a pass of this workload is committed, else the formula
            # charged only for the records of vehicles that drove (waiting / removed slots are neither rewritten nor re-read
            # from HBM: reset image); the nominal formula charges all V records read + written
            moved, moved_src = None, None
            if traffic:
                moved, moved_src = float(traffic), "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (%s)" % traffic_src
            elif work and "driving_traffic_mean" in work and dom == "k_step":
                moved = (b_step - 2 * 128 * (args.traffic - work["driving_traffic_mean"])) * N
                moved_src = "algorithmic bytes charged for the records of driving vehicles only (no counter pass of this workload committed)"
            elif work and "present_agents_mean" in work and dom == "k_step":
                moved = (b_step - 2 * 128 * (A - work["present_agents_mean"])) * N
                moved_src = "algorithmic bytes charged for the records of present agents only (no counter pass of this workload committed)"
            achieved = (moved / (dom_ms * 1e-3) / 1e9) if (moved and dom_ms > 0) else nominal
            out["roofline"] = {
                "bound": "hbm", "kernel": dom + (" (observation fused)" if fused else ""), "achieved": achieved,
                "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "frac_source": moved_src or "nominal algorithmic bytes",
                "bytes_per_launch": moved if moved else dom_bytes,
                "traffic": traffic, "traffic_source": traffic_src,
                # the nominal formula (every record read + written, whether it moved or not)
                "achieved_nominal": nominal, "frac_nominal": nominal / 8000.0,
                "bytes_per_env_step": {"k_step": b_step, "k_observe": b_obs},
                **({"bytes_per_env_step_active": b_step - 2 * 128 * (args.traffic - work["driving_traffic_mean"]),
                    "frac_active": (b_step - 2 * 128 * (args.traffic - work["driving_traffic_mean"])) * N / (dom_ms * 1e-3) / 8e12}
                   if (work and "driving_traffic_mean" in work and dom == "k_step" and dom_ms > 0) else {}),
                "k_step_ms": prof["k_step_ms"], "k_observe_ms": prof["k_observe_ms"], "events": prof["count"],
                "launches_per_event_group": stride if fused else 1,
            }
        else:
            out["roofline"] = None
        if with_cpu_baseline and not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(descs, args) if args.workload == "c3" else None
        else:
            out["cpu_baseline"] = None
    if gatherer is not None:
        gatherer.close()
    for ej in extra:
        ej.close()
    eng.close()
    del actions, act_buf
    torch.cuda.empty_cache()
    return out


def row_summary(name, line):
    """The part of a row's line that goes into `rows` of the headline."""
    r = line.get("roofline") or {}
    c = line["config"]
    keep = ("driving_traffic_mean", "envs_with_traffic_frac", "ego_speed_kmh_mean", "episode_step_mean", "active_agents_mean",
            "present_agents_mean", "step_kernel", "obs_dim", "envs_per_gpu")
    return {
        "row": name, "workload": c["workload"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
        "steps_timed": line["steps_timed"], "warmup_run": line["warmup_run"],
        **{k: c[k] for k in keep if k in c},
        "roofline": {k: r.get(k) for k in ("kernel", "achieved", "frac", "frac_source", "frac_nominal", "frac_active", "traffic",
                                           "traffic_source", "bytes_per_launch", "k_step_ms", "k_observe_ms")} if r else None,
    }


def run_rank(args, rank, world, local_rank):
    import copy
    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise RuntimeError("bench.py needs a GPU: the step engine has no CPU path")
    if world > n_dev and not args.oversubscribe:
        raise RuntimeError("%d ranks but only %d GPUs visible (use --oversubscribe --backend gloo for plumbing tests)" % (world, n_dev))
    torch.cuda.set_device(local_rank % n_dev)
    dev = torch.device("cuda", local_rank % n_dev)
    if world > 1:
        kw = dict(device_id=dev) if args.backend == "nccl" else {}  # binds the RCCL communicator to this rank's GPU
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world, **kw)

    out = measure(args, rank, world, local_rank)
    # the loaded rows, in the same invocation (N = 1, the default command only: any flag that changes the workload of the
    # headline -- other actions, env counts, groups ... -- is a single-workload run)
    default_cmd = (args.workload == "c3" and args.actions == "uniform" and args.traffic_mode == "trigger" and args.envs == 4096 and
                   args.traffic == 16 and args.lasers == 240 and args.maps == 100 and args.groups == 1 and args.engines == 1 and
                   args.step_n == 1 and not args.topdown and not args.exact)
    if world == 1 and rank == 0 and not args.no_rows and (default_cmd or args.rows):
        want = None if not args.rows else set(args.rows.split(","))
        rows = []
        for name, over in ROWS:
            if want is not None and name not in want:
                continue
            ra = copy.copy(args)
            ra.exact, ra.warmup, ra.steps = True, 1500, 2048
            for k, v in over.items():
                setattr(ra, k, v)
            try:
                rows.append(row_summary(name, measure(ra, 0, 1, local_rank, with_cpu_baseline=False)))
            except Exception as ex:  # noqa: BLE001  (a row that fails must not take the metric's line with it)
                rows.append({"row": name, "error": "%s: %s" % (type(ex).__name__, str(ex)[:200])})
        out["rows"] = rows
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _spawned(local_rank, args, world, port):
    os.environ["RANK"] = str(local_rank)
    os.environ["LOCAL_RANK"] = str(local_rank)
    os.environ["WORLD_SIZE"] = str(world)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    run_rank(args, local_rank, world, local_rank)


def main(argv=None):
    args = parse_args(argv)
    # the CPU baseline's OpenMP threads stay where they start (set before libgomp is loaded by liborc.so / torch)
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) >= 1 and "RANK" in os.environ:
        # launched by torch.distributed.run (or any launcher that sets the rendezvous environment)
        world = int(os.environ["WORLD_SIZE"])
        if args.gpus != world and int(os.environ["RANK"]) == 0:
            print("bench.py: --gpus %d but the launcher started %d ranks; reporting n_gpus = %d" % (args.gpus, world, world),
                  file=sys.stderr)
        run_rank(args, int(os.environ["RANK"]), world, int(os.environ.get("LOCAL_RANK", "0")))
        return
    if args.gpus <= 1:
        run_rank(args, 0, 1, 0)
        return
    # `python bench.py --gpus N` without a launcher: start the N ranks here, one per GPU
    import torch
    import torch.multiprocessing as mp
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not args.oversubscribe:
        raise SystemExit("bench.py --gpus %d: only %d GPUs visible" % (args.gpus, n_dev))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_spawned, args=(args, args.gpus, port), nprocs=args.gpus, join=True)


if __name__ == "__main__":
    main()
