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
short run would report an early-episode number.  The bench therefore ALWAYS pre-rolls and times at least the floors, whatever
--warmup / --steps say: at N = 1 BASELINE.md section 4 to the letter -- 10 k warm-up steps, then three windows of 100 k timed steps,
the median is the value (5.5 s of stepping) --, at N > 1 1500 + 3 x 4096 per pass (every transport gets its own windows); the JSON
line carries the requested counts ("steps", "warmup") and the counts actually run ("steps_timed", "warmup_run"); `ms_per_step` and
`value` refer to the timed steps of the median window.  --exact turns the floors off.

Rows.  At N = 1 the same invocation also times the LOADED workloads next to the metric's (whose uniform(-1,1) stream leaves the
ego crawling and most traffic parked): the scripted lane-keeping ego, respawn-mode traffic (every IDM vehicle drives), BASELINE
config 5 (multi-agent roundabout, 8 agents, 240 and 72 beams; 40 agents = the reference's default), and 32768 envs on the one GPU
(config 4's per-node size) -- printed as `rows: [...]`, each with its own roofline (shorter windows: 1500 + 2048 steps;
--no-rows skips them).  `roofline.frac` is charged for the bytes that MOVE: the HBM bytes the rocprofv3 counters saw for that
row (profiles/r*_pmc_*.json, matched by workload) or, without a counter file, the record bytes of the vehicles that drove
(`frac_active`); the nominal formula that charges all V records read + written stays next to it as `frac_nominal`.

Windows.  BASELINE.md section 4 asks for the median of three timed runs: the timed region is --windows (3) consecutive windows of
the timed step count each, every one bracketed by barrier + synchronize; `value` / `ms_per_step` are the MEDIAN window's,
`windows` lists all of them and `window_spread` = (max - min) / median.

N > 1.  Environments are independent: the step itself has no exchange.  The north star adds ONE gather of
(obs, reward, done) per step over xGMI.  One invocation measures, back to back on the same engine state: the step without any
exchange (`value_replicas`: a data-parallel learner that consumes its own shard) and the step WITH the per-step gather under
EVERY transport (`value_by_transport`): "root" = one RCCL gather to rank 0 per step (grouped point-to-point: one direct xGMI link
per peer), eager and captured in a HIP graph per 64-step action cycle ("root+graph"); "peer+graph" / "peer" = direct peer writes
over HIP IPC (pgdrive_amd/peer.py); "collective" / "collective+graph" = RCCL all_gather_into_tensor (every rank receives every
row).  Each pass is followed by the exchange's self-check (`gather_ok`).  `value` = the best transport whose self-check passed,
named in `config.parallelism`; --transport NAME [--graph] runs that one only.  A pass that hangs (a transport that has never run
on this hardware) is cut off by a watchdog after --transport-timeout seconds: rank 0 then prints the line with what was measured
up to that point (`aborted` says where) and every rank exits.
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
PREROLL_MIN = 1500    # steps before the timed region (steady-state traffic): N > 1 (every transport gets its own windows)
TIMED_MIN = 4096      # timed steps per window (>= 64 event groups of PROF_STRIDE launches): N > 1
PREROLL_HEAD = 10000  # N = 1, the metric's row: BASELINE.md section 4 to the letter -- "10 k warm-up, 100 k timed steps, median of 3"
TIMED_HEAD = 100000   #   (three windows of 100 k steps: 5.3 s)


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


PROFILES_DIR = os.path.join(ROOT, "profiles")


def library_sha():
    """The loaded engine library's stamp (pgd_source_sha: sha256 over the sources it was compiled from, pgdrive_amd/build.py); an
    experimental build outside build.py says "unstamped" and matches no committed profile."""
    try:
        from pgdrive_amd import engine
        return engine.load_library().pgd_source_sha().decode()
    except Exception:  # noqa: BLE001  (an older PGD_LIB build without the entry point)
        return "unstamped"


def _match_profile(pattern, N, args):
    """Newest profiles/<pattern> whose recorded workload equals this run's (envs / traffic / beams / actions / traffic mode /
    workload / agents).  The caller compares the record's `source_sha` with the loaded library's before it quotes a figure."""
    import glob
    for p in sorted(glob.glob(os.path.join(PROFILES_DIR, pattern)), reverse=True):
        try:
            t = json.load(open(p))
        except Exception:
            continue
        if (t.get("envs"), t.get("traffic"), t.get("lasers"), t.get("actions", "uniform"), t.get("traffic_mode", "trigger"),
                t.get("workload", "c3"), t.get("agents", 1)) == \
                (N, args.traffic, args.lasers, args.actions, args.traffic_mode, args.workload, args.agents if args.workload == "c5" else 1):
            return t, os.path.basename(p)
    return None, None


def load_traffic(N, args):
    """HBM bytes per k_step launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected in separate
    runs of this same command, newest profiles/r*_pmc_traffic.json whose workload matches); null when none matches."""
    t, src = _match_profile("r*_pmc_traffic.json", N, args)
    if t is None:
        return None, None, None
    return t.get("bytes_per_launch"), src, t.get("bytes_per_launch_k_observe")


def issue_model():
    """ns a gfx950 SIMD spends per wave-instruction it retires, by waves per SIMD and instruction mix: profiles/r*_issue_rate.json,
    the summary of tools/issue_rate.hip run on the GPU box (committed; the microbenchmark itself is not part of a bench run)."""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_issue_rate.json")), reverse=True):
        try:
            return json.load(open(p)), os.path.basename(p)
        except Exception:
            continue
    return None, None


def make_roofline(prof, stride, work, args, N, A, D, lib_sha=None):
    """The `roofline` object of a line.  HBM: algorithmic bytes of the dominant kernel / its mean launch time from the HIP events
    (`frac`, the contract's figure) and the bytes the counters saw move (`frac_moved`).  `issue`: the instruction-issue bound of the
    same kernel from its committed SQ_INSTS_* pass and the measured issue rates of tools/issue_rate.hip.  `bound` names what the
    measurements say limits the kernel."""
    if prof is None:
        return None
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
    # Counter figures are only quoted for the binary they were measured on: every profile summary carries the source stamp of the
    # library that ran under rocprofv3 (tools/pmc_traffic.py, tools/pmc_insts.py); a pass of another build is reported as stale --
    # `traffic` / `issue` null -- instead of describing a kernel that is no longer the one timed here
    lib_sha = lib_sha if lib_sha is not None else library_sha()
    insts, insts_src = _match_profile("r*_pmc_insts.json", N, args)
    t_rec, _ = _match_profile("r*_pmc_traffic.json", N, args)
    prof_sha = (t_rec or {}).get("source_sha") if t_rec else None
    insts_sha = (insts or {}).get("source_sha") if insts else None
    stale_why = []
    if t_rec is not None and prof_sha != lib_sha:
        stale_why.append("%s was measured on source %s, the loaded library is %s" % (traffic_src, prof_sha or "(unstamped: before round 6)", lib_sha))
        traffic, traffic_obs = None, None
    if insts is not None and insts_sha != lib_sha:
        stale_why.append("%s was measured on source %s, the loaded library is %s" % (insts_src, insts_sha or "(unstamped: before round 6)", lib_sha))
        insts = None
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
    achieved_moved = (moved / (dom_ms * 1e-3) / 1e9) if (moved and dom_ms > 0) else None
    # instruction issue: instructions per wave of the dominant kernel (committed SQ_INSTS_* pass of this workload) x waves per
    # SIMD x the time a SIMD needs per wave-instruction at that occupancy and VALU : SALU mix (tools/issue_rate.hip)
    issue = None
    model, model_src = issue_model()
    if insts and model and dom_ms > 0:
        kk = insts.get("k_observe" if dom == "k_observe" else "k_step") or {}
        per_wave = kk.get("insts_per_wave")
        waves = kk.get("waves_per_launch")
        if per_wave and waves:
            wps = waves / 1024.0  # 256 CUs x 4 SIMDs; every wave of these launches is resident at once up to the VGPR limit
            occ = max(1, min(8, int(round(min(wps, kk.get("max_waves_per_simd", 8))))))
            ratio = kk.get("valu", 0) / max(1.0, kk.get("salu", 1))
            row = min(model["mixes"], key=lambda m: abs(m["valu_per_salu"] - ratio) if m.get("valu_per_salu") else 1e9)
            ns = row["ns_per_inst_per_simd"].get(str(occ)) or row["ns_per_inst_per_simd"][max(row["ns_per_inst_per_simd"])]
            bound_us = wps * per_wave * ns * 1e-3
            issue = {"insts_per_wave": per_wave, "valu": kk.get("valu"), "salu": kk.get("salu"), "lds": kk.get("lds"),
                     "vmem": kk.get("vmem"), "smem": kk.get("smem"), "waves_per_simd": wps, "ns_per_inst_per_simd": ns,
                     "cycles_per_inst": ns * model.get("clock_ghz", 2.4), "mix_row": row["name"],
                     "bound_us": bound_us, "kernel_us": dom_ms * 1e3, "frac": bound_us / (dom_ms * 1e3),
                     "source": "%s (SQ_INSTS_* per wave) x %s (measured issue rates)" % (insts_src, model_src)}
    hbm_frac = nominal / 8000.0
    if issue is not None:
        bound = "issue" if issue["frac"] >= 0.7 else ("hbm" if hbm_frac >= 0.7 else "latency")
        note = ("neither pipe is the limit: %.0f %% of the HBM roofline, %.0f %% of the measured issue bound -- the launch is bound by the "
                "dependent chain (loads, LDS round trips, exec-dependent VALU) of its slowest waves plus the launch ramp" % (
                    100 * hbm_frac, 100 * issue["frac"])) if bound == "latency" else None
    else:
        bound, note = "hbm", ("the committed counter passes of this workload belong to another build (stale): only the HBM figure from "
                              "this run's own HIP events is available" if stale_why else
                              "no committed instruction pass of this workload: only the HBM figure is available")
    return {
        "bound": bound, **({"bound_note": note} if note else {}),
        "kernel": dom + (" (observation fused)" if fused else ""),
        # the contract's figure: ALGORITHMIC bytes per launch / mean launch duration
        "achieved": nominal, "peak": 8000.0, "unit": "GB/s", "frac": hbm_frac,
        "bytes_per_launch": dom_bytes, "bytes_per_env_step": {"k_step": b_step, "k_observe": b_obs},
        # what the counters saw move (unchanged records are neither rewritten nor re-read from HBM)
        "traffic": traffic, "traffic_source": traffic_src,
        "achieved_moved": achieved_moved, "frac_moved": (achieved_moved / 8000.0) if achieved_moved else None, "moved_source": moved_src,
        "bytes_per_launch_moved": moved,
        "achieved_nominal": nominal, "frac_nominal": hbm_frac,  # (names of rounds 3 - 4, same numbers as achieved / frac)
        **({"bytes_per_env_step_active": b_step - 2 * 128 * (args.traffic - work["driving_traffic_mean"]),
            "frac_active": (b_step - 2 * 128 * (args.traffic - work["driving_traffic_mean"])) * N / (dom_ms * 1e-3) / 8e12}
           if (work and "driving_traffic_mean" in work and dom == "k_step" and dom_ms > 0) else {}),
        "issue": issue,
        # the binary the counter passes belong to against the binary timed here (equal, or `stale` says why the figures are null)
        "source_sha": lib_sha, "profile_source_sha": prof_sha if t_rec is not None else insts_sha,
        "stale": bool(stale_why), **({"stale_reason": "; ".join(stale_why)} if stale_why else {}),
        "k_step_ms": prof["k_step_ms"], "k_observe_ms": prof["k_observe_ms"], "events": prof["count"],
        "launches_per_event_group": stride if fused else 1,
    }


class Watchdog:
    """N > 1: a transport that hangs (RCCL or HIP IPC between devices that have never met) must not cost the whole line.  Every
    rank arms a timer per transport pass; when it fires, rank 0 prints the line built from what was measured before the pass
    (`partial`, with `aborted`) and every rank leaves with os._exit(0) -- no clean-up of a communicator that no longer answers."""
    def __init__(self):
        self.timer, self.partial, self.rank, self.label = None, None, 0, "the run"

    def arm(self, label, seconds):
        import threading
        self.disarm()
        self.timer = threading.Timer(seconds, self._fire, args=(label, seconds))
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def on_termination(self):
        """A rank that dies hard (a memory fault under a transport that has never run between these devices) makes the launcher
        send SIGTERM to the others.  Rank 0 then still prints what it has: Python's C-level handler writes the signal number to a
        wake-up descriptor at once, even while the main thread sits inside a HIP / RCCL call, and a helper thread that reads the
        descriptor prints the partial line and leaves."""
        import signal
        import threading
        r, w = os.pipe()
        os.set_blocking(w, False)
        signal.signal(signal.SIGTERM, lambda *_: None)  # (installs the C-level handler that feeds the wake-up descriptor)
        signal.set_wakeup_fd(w, warn_on_full_buffer=False)

        def watch():
            os.read(r, 1)
            sys.stderr.write("bench.py: rank %d: SIGTERM (another rank ended?) -- printing what was measured\n" % self.rank)
            if self.rank == 0 and self.partial is not None:
                line = dict(self.partial)
                line["aborted"] = "terminated by the launcher (SIGTERM: another rank ended?) during " + self.label
                print(json.dumps(line), flush=True)
            os._exit(1)
        t = threading.Thread(target=watch, daemon=True)
        t.start()

    def _fire(self, label, seconds):
        sys.stderr.write("bench.py: rank %d: transport %s did not finish within %.0f s -- ending the run\n" % (self.rank, label, seconds))
        sys.stderr.flush()
        if self.rank == 0 and self.partial is not None:
            line = dict(self.partial)
            line["aborted"] = "transport %s exceeded %.0f s" % (label, seconds)
            print(json.dumps(line), flush=True)
        time.sleep(0.5)
        os._exit(0)


WATCHDOG = Watchdog()


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
    ap.add_argument("--transport", default="all", choices=["all", "root", "collective", "peer"],
                    help="the per-step gather.  all (default): every transport back to back in this one invocation -- RCCL gather to rank 0 "
                         "eager and in a HIP graph, direct peer writes over HIP IPC in a HIP graph and eager, RCCL all_gather_into_tensor "
                         "eager and in a HIP graph -- `value` = the best one whose self-check passed.  A name: that transport only "
                         "(eager, or captured with --graph)")
    ap.add_argument("--transport-timeout", type=float, default=150.0,
                    help="N > 1: seconds one transport's pass may take before the watchdog prints the line measured so far and ends the run")
    ap.add_argument("--windows", type=int, default=3,
                    help="timed windows of the timed step count each; value = the median window (BASELINE.md section 4: median of 3)")
    ap.add_argument("--actions", default="uniform", choices=["uniform", "straight", "straight-noise", "expert"],
                    help="uniform(-1,1) (the metric's stream), straight: constant [0, 1] (BASELINE.md section 4, profile_pgdrive.py:16), "
                         "straight-noise: the same with a little steering noise so that episodes differ, or "
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
    ap.add_argument("--groups-graph", dest="groups_graph", type=int, default=0,
                    help="with --groups G: capture U steps of every group in a HIP graph on the group's stream (one graph per group) and "
                         "replay the graphs every U steps -- G host calls per U steps instead of 2 launches x G per step")
    ap.add_argument("--topdown", action="store_true",
                    help="c3 with the top-down image observation (TopDownPGDriveEnv: 84 x 84 x 5, lidar off): pgd_step + "
                         "pgd_observe_topdown per step; reported next to the metric, never as the metric")
    ap.add_argument("--topdown-u8", dest="topdown_u8", action="store_true",
                    help="with --topdown: the image as bytes in [0, 255] (the reference's rgb_clip=False; pgd_observe_topdown_u8)")
    ap.add_argument("--jit", action="store_true",
                    help="build and load a step kernel with this run's configuration compiled in (pgdrive_amd/jit.py, Engine.specialise): for "
                         "configurations the library has no instantiation for; reported as rows, never as the metric")
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
    args.windows = max(1, args.windows)
    return args


# The rows timed next to the metric's workload at N = 1 (name, overrides of the command line).  Pre-roll 1500 steps (C5: 1000, the
# roundabout fills up over the first thousand steps), then --windows (3) windows of `steps` (default 1024) each, median reported.
ROWS = [
    ("c3_straight", dict(actions="straight")),                       # "a second run with constant [0, 1]"
    ("c2_1024", dict(envs=1024, traffic=0, lasers=0, steps=4096)),   # BASELINE config 2: dynamics + localisation + reward only
    ("c3_expert", dict(actions="expert")),
    ("c3_respawn", dict(traffic_mode="respawn")),
    ("c3_expert_respawn", dict(actions="expert", traffic_mode="respawn")),
    # (the multi-agent rows: windows of 2000 steps = two whole 1000-step agent horizons, so that the three windows see the same mix of
    # the population's phases -- with 1024-step windows they differed by 10 - 12 %)
    # (and a pre-roll of 6000 steps: with 1000 the three windows still FELL -- 83.1 / 79.4 / 77.2 M at 40 seats: the roundabout's
    # population was still growing -- and the median overstated the steady state; round 6)
    ("c5_8x240", dict(workload="c5", agents=8, lasers=240, warmup=6000, steps=2000)),
    ("c5_8x72", dict(workload="c5", agents=8, lasers=72, warmup=6000, steps=2000)),
    ("c5_40x72", dict(workload="c5", agents=40, lasers=72, warmup=6000, steps=2000)),
    # the same 4096 envs as two asynchronous env groups of 2048 (pgd_set_groups / pgd_step_group, each on its own stream): the step
    # kernel of one group -- one wave per env, waiting for memory half of its life -- runs beside the observation kernel of the other
    # (a HIP graph of 64 steps per group -- the whole cycle of the action ring, the same action sequence as the eager launches: as eager
    # launches the row needs four launches from the host inside 43 us, which one GPU box of the round's measurements did not manage --
    # 56 M there against 94 M; `--groups 2` without `--groups-graph` is the eager form)
    ("c5_40x72_two_groups", dict(workload="c5", agents=40, lasers=72, warmup=6000, steps=2048, groups=2, groups_graph=64)),
    ("c3_32768", dict(envs=32768, warmup=1500, steps=512)),
    # the top-down image observation (TopDownPGDriveEnv: 84 x 84 x 5 floats per env and step instead of the 274-float row): pgd_step
    # without the lidar + pgd_observe_topdown; a write-bound kernel of its own
    ("c3_topdown", dict(topdown=True, lasers=0, warmup=600, steps=512)),
    # the same image as bytes (the reference's rgb_clip=False: pgd_observe_topdown_u8): a quarter of the float image's writes
    ("c3_topdown_u8", dict(topdown=True, topdown_u8=True, lasers=0, warmup=600, steps=512)),
    # a configuration the library has no instantiation for (72 beams, 12 traffic slots): the general kernel, and the same engine with a
    # step kernel built for it at run time (pgdrive_amd/jit.py; round 6)
    ("c3_72x12_general", dict(lasers=72, traffic=12)),
    ("c3_72x12_run_time_kernel", dict(lasers=72, traffic=12, jit=True)),
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
    jit_s = None
    if getattr(args, "jit", False):  # (2 s of hipcc the first time, cached by configuration)
        t_j = time.perf_counter()
        jit_ok = eng.specialise(wait=True)
        jit_s = (time.perf_counter() - t_j) if jit_ok else None
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
    else:  # "drive straight" (profile_pgdrive.py:16): full throttle; straight-noise: a little steering noise so that episodes differ
        acts = np.zeros((CYC, N, A, 2), dtype=np.float32)
        if args.actions == "straight-noise":
            acts[..., 0] = rng.normal(0, 0.05, size=(CYC, N, A))
        acts[..., 1] = 1.0
    actions = torch.from_numpy(acts).to(dev)

    warm = args.warmup if args.exact else max(args.warmup, PREROLL_HEAD if world == 1 else PREROLL_MIN)
    timed = args.steps if args.exact else max(args.steps, TIMED_HEAD if world == 1 else TIMED_MIN)
    n_win = args.windows
    want_replicas = world == 1 or args.mode in ("both", "replicas")
    want_gather = world > 1 and args.mode in ("both", "gather")
    # N > 1: the replicas pass first (no collective inside its timed loop), then the passes with the per-step gather, one per
    # transport -- a transport that fails on hardware it has never run on leaves the line with the others and says why
    if args.transport == "all":
        # (the RCCL transports first: should the peer transport die hard between devices it has never met, the launcher's SIGTERM
        # still finds a partial line with everything measured before it: Watchdog.on_termination)
        plan = [("root", False), ("root", True), ("collective", False), ("collective", True), ("peer", True), ("peer", False)]
    else:
        plan = [(args.transport, bool(args.graph))]

    if args.groups > 1:
        eng.set_groups(args.groups)
    if args.topdown:
        eng.enable_topdown(uint8=bool(getattr(args, "topdown_u8", False)))

    expert = args.actions == "expert"
    if expert and (A != 1 or args.groups > 1 or args.step_n > 1 or args.engines > 1 or args.topdown or want_gather):
        raise SystemExit("--actions expert: single-agent closed loop on the rank's own shard only (N > 1: --mode replicas)")
    act_buf = torch.zeros((N, A, 2), dtype=torch.float32, device=dev)

    def step_replica(k):
        if expert:  # closed loop: observation of step k - 1 -> scripted policy -> step k, ONE launch (pgd_step_lane_keep; rounds 3 - 5
            eng.step_lane_keep(k)  # launched the policy as a kernel of its own: 4.9 us of a 23.5 us iteration)
            return
        if args.groups > 1:
            if state.get("group_graphs"):  # a single-stream graph of U steps per group, replayed on the group's stream
                if k % args.groups_graph == 0:
                    for g in range(args.groups):
                        with torch.cuda.stream(eng.group_streams[g]):
                            state["group_graphs"][g].replay()
                return
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

    def fence():
        for g in range(args.groups if args.groups > 1 else 0):
            eng.group_sync(g)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor(x, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    def timed_windows(enqueue_window):
        """n_win windows of `timed` steps, each bracketed by barrier + synchronize; `enqueue_window()` enqueues one window's steps
        and returns once they are enqueued (and, for the gather, drained).  Returns the per-window times (max over ranks) and
        the host's enqueue time per step of the median window."""
        wins, enq = [], []
        for w in range(n_win):
            fence()
            t0 = time.perf_counter()
            enqueue_window()
            t_enq = time.perf_counter()  # every launch of the window has been enqueued: the host's share of the loop
            fence()
            t1 = time.perf_counter()
            wins.append(t1 - t0)
            enq.append(t_enq - t0)
        wins = max_over_ranks(wins)
        med = int(np.argsort(wins)[len(wins) // 2])
        return wins, med, enq[med]

    results = {}        # "replicas" -> dict(elapsed, windows, prof, host_enqueue_s)
    by_transport = {}   # "root", "root+graph", ... -> dict (value filled in by make_line)
    state = dict(counter=0, work=None, aborted=None)
    watchdog = WATCHDOG if world > 1 else None

    def make_line(by_t, work, aborted):
        """The JSON line from what has been measured so far (rank 0; also called by the watchdog with work = None)."""
        per_step_units = float(N) * world * max(1, args.engines)
        rate = lambda el: per_step_units * timed / el  # noqa: E731
        k_us = ((results.get("replicas") or {}).get("prof") or {}).get("k_step_ms", 0.0) * 1e3 or None
        slice_bytes = N * pdist.pack_width(D, A) * 4
        link_us = slice_bytes / (XGMI_LINK_GBPS * 1e9) * 1e6
        ring_us = (world - 1) * slice_bytes / (XGMI_LINK_GBPS * 1e9) * 1e6
        table = {}
        for name, r in by_t.items():
            e = dict(transport=r["transport"], hip_graph=r["hip_graph"], describe=r.get("describe"), gather_mem=r.get("gather_mem"))
            if "elapsed" in r:
                host_us = r["host_enqueue_s"] / timed * 1e6
                bound_us = ring_us if r["transport"] == "collective" else link_us
                cands = [x for x in (k_us, bound_us, host_us) if x]
                e.update(value=rate(r["elapsed"]), ms_per_step=r["elapsed"] / timed * 1e3,
                         windows=[rate(w) for w in r["windows"]], gather_ok=r["gather_ok"], gather_check=r["gather_check"],
                         host_enqueue_us_per_step=host_us, hip_graph_steps_per_replay=r["graph_cycle"] or None,
                         xgmi_bound_us_per_step=bound_us,
                         model_floor_us_per_step=max(cands) if cands else None,
                         model_ceiling_env_steps_per_s=per_step_units / (max(cands) * 1e-6) if cands else None)
            if "error" in r:
                e["error"] = r["error"]
            table[name] = e
        # (a peer pass forced onto coarse-grained memory -- PGD_GATHER_COARSE=1 -- is an A/B figure, never the headline: ADVICE r05)
        good = [n for n, e in table.items() if e.get("gather_ok") and "error" not in e and e.get("gather_mem") != "coarse"]
        measured = [n for n, e in table.items() if "value" in e]
        best = max(good, key=lambda n: table[n]["value"]) if good else None
        # the headline: the best transport whose self-check passed; without one, a measured transport (gather_ok false says so);
        # without any, the replicas pass
        head = best or (max(measured, key=lambda n: table[n]["value"]) if measured else None)
        if head is not None:
            elapsed, wins = by_t[head]["elapsed"], by_t[head]["windows"]
        elif "replicas" in results:
            elapsed, wins = results["replicas"]["elapsed"], results["replicas"]["windows"]
        else:
            raise RuntimeError("nothing was measured: " + "; ".join("%s: %s" % (n, e.get("error")) for n, e in table.items()))
        out = {
            "metric": "env-steps/sec (whole node) at 4096 envs x 240 lidar beams",
            "value": rate(elapsed), "unit": "env-steps/s", "n_gpus": state.get("ranks_ran", world), "steps": args.steps, "warmup": args.warmup,
            "steps_timed": timed, "warmup_run": warm, "steps_effective": timed, "rccl_ranks": state.get("rccl_ranks"),
            "ms_per_step": elapsed / timed * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "windows": [rate(w) for w in wins], "window_spread": (max(wins) - min(wins)) / elapsed if elapsed > 0 else None,
            "window_rule": "median of %d windows of %d steps" % (n_win, timed),
        }
        if aborted:
            out["aborted"] = aborted
        if world == 1 and head is None and "replicas" in results:  # the host's share of a step: launches enqueued, not yet run
            out["host_enqueue_us_per_step"] = results["replicas"]["host_enqueue_s"] / timed * 1e6
        if world > 1:
            out["value_mode"] = ("gather:" + head) if head is not None else "replicas"
            if "replicas" in results:
                out["value_replicas"] = rate(results["replicas"]["elapsed"])
                out["ms_per_step_replicas"] = results["replicas"]["elapsed"] / timed * 1e3
                out["windows_replicas"] = [rate(w) for w in results["replicas"]["windows"]]
            if table:
                out["value_by_transport"] = table
                out["transport_chosen"] = head
                out["gather_mem"] = next((e["gather_mem"] for e in table.values() if e.get("gather_mem")), None)
            if head is not None:
                out["value_gather"] = table[head]["value"]
                out["gather_ok"] = table[head]["gather_ok"]
                out["gather_check"] = table[head]["gather_check"]
                # what the exchange can cost by construction: each peer's slice over its own xGMI link into
                # the root (transport root / peer: the links work in parallel, the slowest is one slice), the all-gather as a ring
                # (per-link bound: (n - 1) slices through every link), and the host's enqueue time per step of this very loop
                out["gather_model"] = {
                    "slice_bytes_per_rank_per_step": slice_bytes, "xgmi_link_GBps": XGMI_LINK_GBPS,
                    "link_bound_us": link_us, "ring_allgather_bound_us": ring_us, "k_step_us": k_us,
                    "host_enqueue_us_per_step": table[head]["host_enqueue_us_per_step"],
                    "host_enqueue_us_per_step_replicas": (results["replicas"]["host_enqueue_s"] / timed * 1e6) if "replicas" in results else None,
                    "hip_graph_steps_per_replay": table[head]["hip_graph_steps_per_replay"],
                    "predicted_floor_us_per_step": table[head]["model_floor_us_per_step"],
                    "predicted_ceiling_env_steps_per_s": table[head]["model_ceiling_env_steps_per_s"],
                    "note": "the exchange of step t overlaps the kernels of step t + 1 (double-buffered): the step rate is bounded by the "
                            "slowest of kernel, link and host enqueue, not by their sum",
                }
            elif table:
                out["gather_error"] = "; ".join("%s: %s" % (n, e.get("error")) for n, e in table.items())
        if head is not None:
            par = "env-sharded dp%d + %s, double-buffered" % (world, table[head]["describe"])
        else:
            par = "env-sharded dp%d, no data-path collective" % world
        c2 = args.workload == "c3" and args.traffic == 0 and args.lasers == 0
        TD_BYTES = 1 if getattr(args, "topdown_u8", False) else 4
        out["config"] = {
            "workload": (("C2: %d envs/GPU x 1 ego, no traffic, no lidar (state observation only), PGDrive-v0 maps seeds 1000-1099, "
                          "%s actions, auto-reset" % (N, args.actions)) if c2 else
                         ("C3: %d envs/GPU x (1 ego + %d IDM traffic slots) x %d lidar beams, PGDrive-v0 maps "
                          "seeds 1000-1099, %s actions, auto-reset" % (
                              N, args.traffic, args.lasers, {"uniform": "uniform(-1,1)", "straight": "constant [0, 1] (drive straight)",
                                                             "straight-noise": "drive-straight with steering noise",
                                                             "expert": "scripted lane-keeping (30 km/h)"}[args.actions]) +
                          ("" if args.traffic_mode == "trigger" else ", traffic mode " + args.traffic_mode)))
            if args.workload == "c3" else
            ("C5: %d envs/GPU x %d agents, multi-agent roundabout, %d beams x 40 m, %s actions, respawn, auto-reset; "
             "agent-steps/s = value x %d" % (N, A, args.lasers, args.actions, A)),
            **({"note": "%d independent engines x %d envs on their own streams, stepped round-robin: consecutive steps "
                        "of different engines overlap (asynchronous vector-env groups)" % (args.engines, N)}
               if args.engines > 1 else {}),
            **({"active_vehicles_mean": 1.0 + work["driving_traffic_mean"]} if work and "driving_traffic_mean" in work else {}),
            **(work or {}),
            "step_kernel": state.get("step_kernel"),
            **({"run_time_kernel_build_s": jit_s} if getattr(args, "jit", False) else {}),
            "envs_per_gpu": N * max(1, args.engines), "global_envs": N * world * max(1, args.engines), "obs_dim": D,
            "engines_per_gpu": max(1, args.engines), "env_groups": args.groups,
            **({"env_groups_graph_steps": args.groups_graph} if args.groups > 1 and getattr(args, "groups_graph", 0) > 1 else {}),
            **({"open_loop": "pgd_step_n: %d steps of the action ring per call, one observation per call -- NOT the metric's closed "
                             "loop" % args.step_n} if args.step_n > 1 else {}),
            **({"observation": "top-down image 84 x 84 x 5 %s, %.1f MB written per step" % (
                "uint8 (pgd_observe_topdown_u8)" if TD_BYTES == 1 else "float32 (pgd_observe_topdown)", N * 84 * 84 * 5 * TD_BYTES / 1e6),
                # the image kernel's share of a step = step time - k_step's event time; its roofline is the HBM write rate
                "topdown_us": (elapsed / timed * 1e3 - (((results.get("replicas") or {}).get("prof") or {}).get("k_step_ms") or 0.0)) * 1e3,
                "topdown_write_frac_of_hbm_peak": (N * 84 * 84 * 5 * TD_BYTES) / max(1e-9, (elapsed / timed - (((results.get("replicas") or {}).get("prof") or {}).get("k_step_ms") or 0.0) * 1e-3)) / 8e12}
               if args.topdown else {}),
            "parallelism": par, "backend": (args.backend if world > 1 else "none"),
            "steady_state": "pre-roll %d steps, %d x %d timed steps (floors %d / %d%s)" % (
                warm, n_win, timed, PREROLL_HEAD if world == 1 else PREROLL_MIN, TIMED_HEAD if world == 1 else TIMED_MIN,
                ", off: --exact" if args.exact else ""),
        }
        out["roofline"] = make_roofline((results.get("replicas") or {}).get("prof"), (results.get("replicas") or {}).get("stride", 1),
                                        work, args, N, A, D, lib_sha=state.get("lib_sha"))
        return out
    make_line_ref = [make_line]

    with torch.cuda.stream(eng.stream):
        for k in range(warm):  # pre-roll to steady-state traffic, once, shared by every pass
            step_replica(state["counter"])
            state["counter"] += 1

        if want_replicas:
            for k in range(16 if not args.exact else min(16, args.warmup)):
                step_replica(state["counter"])
                state["counter"] += 1
            # HIP events bracket groups of PROF_STRIDE consecutive k_step launches over the whole timed region; the average
            # launch duration is group time / PROF_STRIDE (two event packets around EVERY launch leave the command processor
            # idle between back-to-back kernels and slow the thing being measured)
            profiled = args.step_n == 1  # (the open-loop variant mixes launches with and without the observation)
            if args.groups > 1 and getattr(args, "groups_graph", 0) > 1:
                # the groups' steps from here on come out of graphs: U steps of group g with the actions of steps 0 .. U - 1 (+ 5 g), captured
                # on the group's stream after the pre-roll (the launches inside a graph carry no event pairs: no k_step event time)
                profiled = False
                fence()
                U = args.groups_graph
                graphs = []
                for g in range(args.groups):
                    gk = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gk, stream=eng.group_streams[g]):
                        for j in range(U):
                            eng.step_group(g, actions[(j + 5 * g) % CYC])
                    graphs.append(gk)
                torch.cuda.synchronize(dev)
                state["group_graphs"] = graphs
                state["counter"] = (state["counter"] + U - 1) // U * U
            # short --exact runs (tests, sweeps of a few hundred steps): smaller groups, so that several of them complete
            stride = PROF_STRIDE if timed >= 8 * PROF_STRIDE else (16 if timed >= 16 else 1)
            if profiled:
                eng.profile_begin(n_win * (timed // stride + 1) + 1, stride=stride)

            def window_replicas():
                for k in range(timed):
                    step_replica(state["counter"])
                    state["counter"] += 1
            wins, med, enq = timed_windows(window_replicas)
            prof = eng.profile_end() if profiled else None
            results["replicas"] = dict(elapsed=wins[med], windows=wins, prof=prof, host_enqueue_s=enq, stride=stride)

        # ---- the per-step gather, one pass per transport (N > 1) -----------------------------------------------------------------
        def run_transport(transport, use_graph):
            res = dict(transport=transport, hip_graph=bool(use_graph))
            g = None
            try:
                if use_graph and transport != "peer" and args.backend != "nccl":
                    raise RuntimeError("backend %s runs its collectives on the host: they cannot be captured in a HIP graph" % args.backend)
                if use_graph and (timed % CYC or CYC % 2):
                    # (a shorter cycle would replay actions[0:cyc] only -- another workload than the eager passes)
                    raise RuntimeError("the timed steps (%d) are not whole %d-step action cycles: no graph pass" % (timed, CYC))
                g = pdist.StepGather(torch, dist, N, D, A, device=dev, transport=transport, engine_lib=eng.L,
                                     device_seq=bool(use_graph and transport == "peer"))
                res["describe"] = g.describe() + (", %d steps per HIP graph replay" % CYC if use_graph else "")
                res["gather_mem"] = g.gather_mem

                def one_step(k):
                    a_k = actions[k % CYC]
                    g.step(lambda rows: eng.step_packed(a_k, rows))
                for k in range(16 if not args.exact else min(16, max(2, args.warmup))):  # buffers / communicator warm-up
                    one_step(state["counter"])
                    state["counter"] += 1
                if g.k % g.nbuf:
                    one_step(state["counter"])
                    state["counter"] += 1
                g.drain()
                fence()
                cycle = None
                if use_graph:
                    cycle = g.capture_cycle([(lambda rows, a_i=actions[i]: eng.step_packed(a_i, rows)) for i in range(CYC)])

                def window_gather():
                    if cycle is not None:  # one host call per cycle of CYC steps
                        for c in range(timed // CYC):
                            cycle.replay()
                    else:
                        for k in range(timed):
                            one_step(state["counter"] + k)
                    state["counter"] += timed
                    g.drain()
                wins, med, enq = timed_windows(window_gather)
                # self-check of the exchange, after the timed loop: every rank's checksum of the rows it produced against the
                # checksum of what arrived for it (a transport that delivers wrong rows would otherwise still print a number)
                a_chk = actions[state["counter"] % CYC]
                corrupt = None
                if args.corrupt_gather:
                    def corrupt(buf):
                        if buf.shape[0] > N or world == 1:  # the rank that holds the gathered rows
                            buf[buf.shape[0] - 1, 3] += 1.0
                ok, detail = g.validate(lambda rows: eng.step_packed(a_chk, rows), corrupt=corrupt)
                state["counter"] += 1
                res.update(elapsed=wins[med], windows=wins, host_enqueue_s=enq, gather_ok=bool(ok), gather_check=dict(ok=bool(ok), **detail),
                           graph_cycle=CYC if cycle is not None else 0)
            except Exception as ex:  # noqa: BLE001  (a transport that does not come up: report it, keep the other passes)
                res["error"] = "%s: %s" % (type(ex).__name__, str(ex)[:300])
                print("bench.py: transport %s%s failed on rank %d: %s" % (transport, "+graph" if use_graph else "", rank, res["error"]),
                      file=sys.stderr, flush=True)
            finally:
                if g is not None:
                    try:
                        g.close()
                    except Exception as ex:  # noqa: BLE001
                        res.setdefault("error", "close: %s: %s" % (type(ex).__name__, str(ex)[:200]))
            return res

        if want_gather:
            for transport, use_graph in plan:
                name = transport + ("+graph" if use_graph else "")
                if watchdog is not None:
                    if rank == 0 and ("replicas" in results or any("elapsed" in r for r in by_transport.values())):
                        watchdog.partial = make_line_ref[0](dict(by_transport), None, None)
                    watchdog.label = "transport " + name
                    watchdog.arm(name, args.transport_timeout)
                by_transport[name] = run_transport(transport, use_graph)
                if watchdog is not None:
                    watchdog.disarm()
                    watchdog.label = "the work statistics after the transports"
                # the ranks agree on what happened (an error on one rank is an error of the pass)
                if world > 1:
                    bad = torch.tensor([1.0 if "error" in by_transport[name] else 0.0], dtype=torch.float64, device=dev)
                    dist.all_reduce(bad)
                    if bad.item() > 0 and "error" not in by_transport[name]:
                        by_transport[name]["error"] = "failed on %d other rank(s)" % int(bad.item())

    step_kernel = eng.describe_step()  # which k_step instantiation the timed launches were (pgd_describe_step)
    state["step_kernel"] = step_kernel
    try:
        state["lib_sha"] = eng.L.pgd_source_sha().decode()  # of the library THIS engine runs on (PGD_LIB builds included)
    except AttributeError:
        state["lib_sha"] = "unstamped"
    # how much work a step does at this point of the run: 5 snapshots of the state, 50 untimed steps apart
    work = None
    if args.workload == "c3" and N <= 65536:
        from pgdrive_amd import _abi as abi
        act_n, with_t, ep = [], [], []
        with torch.cuda.stream(eng.stream):
            for snap in range(5):
                for k in range(50 if snap else 0):
                    step_replica(state["counter"])
                    state["counter"] += 1
                fence()
                f_, i_, ei_ = eng.get_state()
                drv = (i_[abi.SI["STATUS"]][:, A:] == abi.ST_ACTIVE)
                act_n.append(float(drv.sum(axis=1).mean()))
                with_t.append(float(drv.any(axis=1).mean()))
                ep.append(float(ei_[abi.EI["EP_STEPS"]].mean()))
        spd = float(np.abs(f_[abi.SF["SPEED"]][:, 0]).mean() * 3.6)
        work = dict(driving_traffic_mean=float(np.mean(act_n)), envs_with_traffic_frac=float(np.mean(with_t)),
                    episode_step_mean=float(np.mean(ep)), ego_speed_kmh_mean=spd)
        if work["episode_step_mean"] < 5.0:
            # (expert + respawn traffic -- the reference's respawn mode fills every 10 m slot around the ego's spawn
            # point, traffic_manager.py:199-202,292-309, and the scripted ego drives into the vehicle 1 m ahead of it at once)
            work["note"] = ("episodes last %.1f steps on average: this row times reset churn in a jam around the spawn point, not "
                            "driving" % work["episode_step_mean"])
    elif args.workload == "c5":  # (the agent population swings with the 1000-step agent horizon: five snapshots, 100 steps apart)
        from pgdrive_amd import _abi as abi
        act_a, pres_a = [], []
        with torch.cuda.stream(eng.stream):
            for snap in range(5):
                for k in range(100 if snap else 0):
                    step_replica(state["counter"])
                    state["counter"] += 1
                fence()
                f_, i_, ei_ = eng.get_state()
                st_ = i_[abi.SI["STATUS"]][:, :A]
                act_a.append(float((st_ == abi.ST_ACTIVE).sum(axis=1).mean()))
                pres_a.append(float(((st_ == abi.ST_ACTIVE) | (st_ == abi.ST_DYING)).sum(axis=1).mean()))
        work = dict(active_agents_mean=float(np.mean(act_a)), present_agents_mean=float(np.mean(pres_a)),
                    active_agents_snapshots=[round(v, 2) for v in act_a])

    if world > 1:
        if args.backend == "nccl":  # proves that RCCL itself saw every rank (the first multi-GPU run is a first run)
            assert dist.get_backend() == "nccl"
            state["rccl_ranks"] = dist.get_world_size()
        t = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        state["ranks_ran"] = int(round(t.item()))

    out = None
    if rank == 0:
        out = make_line(by_transport, work, None)
        if with_cpu_baseline and not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(descs, args) if (args.workload == "c3" and args.traffic > 0) else None
        else:
            out["cpu_baseline"] = None
    for ej in extra:
        ej.close()
    eng.close()
    del actions, act_buf
    torch.cuda.empty_cache()
    return out


def measure_policy(args, local_rank, steps=2048, warm=1500, windows=3):
    """The closed loop an RL user runs: observation -> policy network -> action -> pgd_step, nothing pre-generated.
    Policy = the 274-256-256-2 tanh MLP of examples/graph_rollout.py (the shape of the reference's PPO expert,
    examples/ppo_expert/numpy_expert.py), random weights (seed 0), C3 workload.  Two implementations of the same network --
      torch         fp32 torch ops (3 x addmm on hipBLASLt + 3 x tanh: six dependent launches per step; the user's unchanged code)
      fused         pgd_mlp_policy: the engine's one-launch MLP (f32 matrix cores, exact f32, activations in LDS: pgdrive_amd/csrc/pgd_policy.h)
      fused_bf16x3  pgd_mlp_policy_prepared: the same launch with split bf16 operands on prepared weights (3 bf16 matrix instructions per product)
    -- and three ways to drive them, same engine configuration:
      eager        one Python iteration per step, 4096 envs
      graph        4 iterations captured in ONE HIP graph (torch.cuda.graphs), replayed: one host call per 4 steps, 4096 envs
      groups_graph two env groups of 4096 (pgd_set_groups / pgd_step_group, 8192 envs in the handle), each group's policy + step
                   captured in its own single-stream HIP graph on the group's stream and replayed side by side: the policy of one
                   group overlaps the step of the other (one graph over both streams costs HIP 4 us of host time per node: measured)
    Reported per variant: env-steps/s (median of `windows` windows), us per iteration, the host's enqueue time per step."""
    import torch
    from pgdrive_amd import _abi, bank, mapdata, scenario
    from pgdrive_amd.engine import Engine
    dev = torch.device("cuda", local_rank)
    descs = bank.get_descriptions(range(1000, 1000 + args.maps))
    mb = mapdata.MapBank(descs)
    sb = scenario.ScenarioBank(descs, [d["seed"] for d in descs], num_agents=1, num_traffic=16, traffic_mode="trigger")
    torch.manual_seed(0)
    lin = [torch.nn.Linear(274, 256), torch.nn.Linear(256, 256), torch.nn.Linear(256, 2)]
    W = [l.weight.detach().t().contiguous().to(dev) for l in lin]
    B = [l.bias.detach().contiguous().to(dev) for l in lin]
    weights = (W[0], B[0], W[1], B[1], W[2], B[2])

    def policy_torch(obs2d, act_out2d):  # tanh MLP; the last activation is written straight into the action buffer the step reads
        h = torch.tanh(torch.addmm(B[0], obs2d, W[0]))
        h = torch.tanh(torch.addmm(B[1], h, W[1]))
        torch.tanh(torch.addmm(B[2], h, W[2]), out=act_out2d)

    def engine(n):
        cfg = _abi.make_config(n, num_agents=1, num_traffic=16, num_lasers=240, auto_reset=1, seed=1234)
        e = Engine(cfg, mb, sb, device=local_rank)
        e.reset(np.arange(n) % len(descs))
        return e

    def timed(run_window, n_env_steps_per_window):
        wins, enq = [], []
        for w in range(windows):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            run_window()
            t1 = time.perf_counter()
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            wins.append(t2 - t0)
            enq.append(t1 - t0)
        med = int(np.argsort(wins)[len(wins) // 2])
        return dict(value=n_env_steps_per_window / wins[med], windows=[n_env_steps_per_window / w for w in wins],
                    window_spread=(max(wins) - min(wins)) / wins[med], us_per_iteration=wins[med] / steps * 1e6,
                    host_enqueue_us_per_step=enq[med] / steps * 1e6)

    N, UNROLL = 4096, 4
    out = {"row": "c3_policy", "workload": "C3 closed loop: %d envs x (1 ego + 16 IDM traffic slots) x 240 beams, actions = tanh MLP 274-256-256-2 "
                                           "(fp32, random weights) of the last observation, auto-reset" % N,
           "unit": "env-steps/s", "steps_timed": steps, "warmup_run": warm,
           "policy": {"torch": "3 x addmm (hipBLASLt) + 3 x tanh per step, fp32", "fused": "pgd_mlp_policy: one launch, f32 MFMA (v_mfma_f32_16x16x4_f32), exact f32",
                      "fused_bf16x3": "pgd_mlp_policy_prepared: one launch, every operand split hi + lo into two bf16, three v_mfma_f32_16x16x32_bf16 "
                                      "per product, f32 accumulation: ~3e-5 on an action (the exact kernel: ~1e-6)"}}
    for impl in ("torch", "fused", "fused_bf16x3"):
        res = {}
        try:  # one engine of 4096 envs: eager, then 4 iterations per HIP graph
            with torch.no_grad():
                eng = engine(N)
                act = torch.zeros((N, 1, 2), dtype=torch.float32, device=dev)
                obs2d, act2d = eng.obs.view(N, -1), act.view(N, 2)

                prep = eng.mlp_prepare(weights) if impl == "fused_bf16x3" else None  # (once per policy update; here: once)

                def iteration():
                    if impl == "torch":
                        policy_torch(obs2d, act2d)
                    else:
                        eng.mlp_policy(weights, act, final_tanh=True, prepared=prep)
                    eng.step(act)
                s = torch.cuda.Stream(device=dev)
                s.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(s):
                    for _ in range(warm):
                        iteration()
                    torch.cuda.synchronize(dev)
                    res["eager"] = timed(lambda: [iteration() for _ in range(steps)], N * steps)
                    torch.cuda.synchronize(dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for _ in range(UNROLL):
                        iteration()
                torch.cuda.synchronize(dev)
                for _ in range(8):
                    g.replay()
                res["graph"] = dict(timed(lambda: [g.replay() for _ in range(steps // UNROLL)], N * steps), steps_per_replay=UNROLL)
                if impl == "torch":
                    f_, i_, _e = eng.get_state()
                    out["ego_speed_kmh_mean"] = float(np.abs(f_[_abi.SF["SPEED"]][:, 0]).mean() * 3.6)
                    out["driving_traffic_mean"] = float((i_[_abi.SI["STATUS"]][:, 1:] == _abi.ST_ACTIVE).sum(axis=1).mean())
                    out["step_kernel"] = eng.describe_step()
                del g
                eng.close()
        except Exception as ex:  # noqa: BLE001
            res["error"] = "%s: %s" % (type(ex).__name__, str(ex)[:300])
        # env groups (pgd_set_groups): a single-stream graph per group, replayed on the group's stream -- 2 x 4096 (the verdict's
        # variant), and smaller groups, which leave register file for the policy's waves next to the other group's step
        for Gn, Ng in {"fused": ((2, N), (2, N // 2), (4, N // 2)), "fused_bf16x3": ((2, N), (2, N // 2))}.get(impl, ((2, N), )):
            key = "groups_graph" if (Gn, Ng) == (2, N) else "groups_graph_%dx%d" % (Gn, Ng)
            try:
                with torch.no_grad():
                    eng = engine(Gn * Ng)
                    eng.set_groups(Gn)
                    act = torch.zeros((Gn * Ng, 1, 2), dtype=torch.float32, device=dev)
                    views = [(eng.obs[eng.group_slice(k)].view(Ng, -1), act[eng.group_slice(k)].view(Ng, 2)) for k in range(Gn)]
                    gs = eng.group_streams
                    prep = eng.mlp_prepare(weights) if impl == "fused_bf16x3" else None
                    eng.sync()

                    def group_iteration(k):  # (on the group's stream)
                        if impl == "torch":
                            policy_torch(*views[k])
                        else:
                            eng.mlp_policy(weights, act, group=k, final_tanh=True, prepared=prep)
                        eng.step_group(k, act)
                    cur = torch.cuda.current_stream(dev)
                    graphs = []
                    for k in range(Gn):
                        gs[k].wait_stream(cur)
                        with torch.cuda.stream(gs[k]):
                            for _ in range(warm):
                                group_iteration(k)
                    torch.cuda.synchronize(dev)
                    for k in range(Gn):
                        gk = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gk, stream=gs[k]):
                            for _ in range(UNROLL):
                                group_iteration(k)
                        graphs.append(gk)
                    torch.cuda.synchronize(dev)

                    def replay_all():
                        for k in range(Gn):
                            with torch.cuda.stream(gs[k]):
                                graphs[k].replay()
                    for _ in range(8):
                        replay_all()
                    res[key] = dict(timed(lambda: [replay_all() for _ in range(steps // UNROLL)], Gn * Ng * steps), steps_per_replay=UNROLL,
                                    env_groups=Gn, envs=Gn * Ng, note="one iteration = every group stepped once (%d env-steps)" % (Gn * Ng))
                    del graphs
                    eng.close()
            except Exception as ex:  # noqa: BLE001
                res[key] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
        out[impl] = res
    vals = [(v["value"], "%s/%s" % (i, k)) for i in ("torch", "fused", "fused_bf16x3") for k, v in (out.get(i) or {}).items()
            if isinstance(v, dict) and "value" in v]
    out["value"], out["value_variant"] = max(vals) if vals else (None, None)
    return out


def row_summary(name, line):
    """The part of a row's line that goes into `rows` of the headline."""
    r = line.get("roofline") or {}
    c = line["config"]
    keep = ("driving_traffic_mean", "envs_with_traffic_frac", "ego_speed_kmh_mean", "episode_step_mean", "active_agents_mean",
            "present_agents_mean", "step_kernel", "obs_dim", "envs_per_gpu", "note", "observation", "topdown_us",
            "topdown_write_frac_of_hbm_peak", "run_time_kernel_build_s", "env_groups", "env_groups_graph_steps")
    iss = r.get("issue") or None
    return {
        "row": name, "workload": c["workload"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
        "windows": line.get("windows"), "window_spread": line.get("window_spread"),
        "steps_timed": line["steps_timed"], "warmup_run": line["warmup_run"], "host_enqueue_us_per_step": line.get("host_enqueue_us_per_step"),
        **{k: c[k] for k in keep if k in c},
        "roofline": {**{k: r.get(k) for k in ("bound", "kernel", "achieved", "frac", "frac_moved", "moved_source", "frac_active", "traffic",
                                              "traffic_source", "bytes_per_launch", "k_step_ms", "k_observe_ms", "source_sha",
                                              "profile_source_sha", "stale")},
                     "issue": ({k: iss.get(k) for k in ("insts_per_wave", "waves_per_simd", "ns_per_inst_per_simd", "bound_us", "frac")}
                               if iss else None)} if r else None,
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

    WATCHDOG.rank = rank
    if world > 1:
        WATCHDOG.on_termination()
    out = measure(args, rank, world, local_rank)
    # the loaded rows, in the same invocation (N = 1, the default command only: any flag that changes the workload of the
    # headline -- other actions, env counts, groups ... -- is a single-workload run)
    default_cmd = (args.workload == "c3" and args.actions == "uniform" and args.traffic_mode == "trigger" and args.envs == 4096 and
                   args.traffic == 16 and args.lasers == 240 and args.maps == 100 and args.groups == 1 and args.engines == 1 and
                   args.step_n == 1 and not args.topdown and not args.exact and not args.jit)
    if world == 1 and rank == 0 and not args.no_rows and (default_cmd or args.rows):
        want = None if not args.rows else set(args.rows.split(","))
        rows = []
        for name, over in ROWS:
            if want is not None and name not in want:
                continue
            ra = copy.copy(args)
            ra.exact, ra.warmup, ra.steps = True, 1500, 1024  # (x --windows windows)
            for k, v in over.items():
                setattr(ra, k, v)
            try:
                rows.append(row_summary(name, measure(ra, 0, 1, local_rank, with_cpu_baseline=False)))
            except Exception as ex:  # noqa: BLE001  (a row that fails must not take the metric's line with it)
                rows.append({"row": name, "error": "%s: %s" % (type(ex).__name__, str(ex)[:200])})
        # the closed loop with a policy network in it (not a `measure` workload: the actions come from the observation)
        if want is None or "c3_policy" in want:
            try:
                rows.append(measure_policy(args, local_rank))
            except Exception as ex:  # noqa: BLE001
                rows.append({"row": "c3_policy", "error": "%s: %s" % (type(ex).__name__, str(ex)[:200])})
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
