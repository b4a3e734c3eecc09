"""The 40-seat roundabout stepped as two asynchronous env groups (double-buffered sampling).

    python examples/marl_env_groups.py --envs 4096 --steps 6000

A multi-agent step is two launches: the step kernel (one wave per env, waiting for memory about half of its life) and the
four-wave observation kernel.  Stepped as ONE batch they run one after the other.  `env.set_groups(2)` splits the same handle
into two halves with their own streams; `step_group(g, actions)` enqueues a half's two launches on its stream and returns, so the
step kernel of one half runs beside the observation kernel of the other (bench.py, rows c5_40x72 / c5_40x72_two_groups: 81 M ->
93 M env-steps/s at 4096 envs).  Envs do not interact, so the halves leave exactly the bytes the single batch leaves
(tests/test_parity_gpu.py::test_multi_agent_env_groups_step_like_one_batch); a learner that consumes group A's rows while group
B steps gets the overlap for free.  Work that reads a group's rows belongs on `engine.group_streams[g]`.

As eager launches the two halves need four kernel launches from the host inside ~43 us per iteration; a slow or busy host turns the
gain into a loss (two of the round's GPU boxes: 53 - 56 M with two groups against 84 - 92 M with one; two others: 96 against 84).
The third line this script prints is the form that does not depend on the host: the steps of each group -- in a rollout: its
policy and its step -- captured in a HIP graph on the group's stream, one host call per group and replay.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout without installing
from pgdrive_amd import MultiAgentRoundaboutVecEnv  # noqa: E402

RING = 16  # pre-drawn action tensors (open loop here; see fused_policy_rollout.py for a policy in the loop)


def rollout(env, steps, groups, throttle, graphs):
    eng = env.engine
    N, A = env.num_envs, env.A
    dev = eng.obs.device
    ret = [torch.zeros(1, device=dev) for _ in range(max(groups, 1))]
    acts = [torch.rand((N, A, 2), device=dev) * 2 - 1 for _ in range(RING)]
    if throttle:
        for a in acts:
            a[..., 1] = a[..., 1] * 0.5 + 0.5  # mostly throttle: the roundabout fills up
    torch.cuda.synchronize()

    def iteration(k):
        # (the consumer below runs every RING-th step only: two torch ops and a stream switch per step cost the host more than the
        # GPU needs for the step)
        if groups <= 1:
            _, rew, _, _ = env.step(acts[k % RING])
            if k % RING == RING - 1:
                ret[0].add_(rew.sum())
            return
        for g in range(groups):
            _, rew, _, _ = env.step_group(g, acts[(k + 5 * g) % RING])  # (the full action tensor: the group reads its own rows)
            if k % RING == RING - 1:
                with torch.cuda.stream(eng.group_streams[g]):  # the consumer of a group's rows runs on the group's stream
                    ret[g].add_(rew.sum())

    def sync_all():
        for g in range(groups if groups > 1 else 0):
            env.group_sync(g)
        torch.cuda.synchronize()

    for k in range(3008):  # pre-roll: the roundabout fills up over the first thousands of steps
        iteration(k)
    sync_all()
    replay = None
    if graphs:  # RING steps of every group (and the consumer of their last rows) in one graph per group
        captured = []
        for g in range(groups):
            gk = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gk, stream=eng.group_streams[g]):
                for j in range(RING):
                    _, rew, _, _ = env.step_group(g, acts[(j + 5 * g) % RING])
                ret[g].add_(rew.sum())
            captured.append(gk)
        torch.cuda.synchronize()

        def replay():
            for g in range(groups):
                with torch.cuda.stream(eng.group_streams[g]):
                    captured[g].replay()
    steps = steps // RING * RING
    t0 = time.perf_counter()
    if replay is not None:
        for _ in range(steps // RING):
            replay()
    else:
        for k in range(steps):
            iteration(k)
    t_enq = time.perf_counter() - t0  # the host's share (with the GPU's queue full it waits for the GPU: an upper bound)
    sync_all()
    dt = time.perf_counter() - t0
    return N * steps / dt, float(sum(r.item() for r in ret)), t_enq / steps * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--throttle", action="store_true", help="mostly-throttle actions instead of uniform(-1, 1)")
    args = ap.parse_args()
    for groups, graphs, what in ((1, False, "one batch, eager launches"), (2, False, "two env groups, eager launches"),
                                 (2, True, "two env groups, a HIP graph of %d steps per group" % RING)):
        env = MultiAgentRoundaboutVecEnv(dict(num_envs=args.envs, num_agents=40, seed=3))
        env.reset()
        if groups > 1:
            env.set_groups(groups)
        rate, ret, host_us = rollout(env, args.steps, groups, args.throttle, graphs)
        status = env.slot_table()[0]
        print("%-52s %6.1f M env-steps/s (%4.0f M agent seats/s), host <= %4.1f us per step, %.1f agents alive per env, reward sum %.0f" %
              (what, rate / 1e6, rate * env.A / 1e6, host_us, float((status == 2).sum(axis=1).mean()), ret))
        env.close()


if __name__ == "__main__":
    main()
