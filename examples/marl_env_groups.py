"""The 40-seat roundabout stepped as two asynchronous env groups (double-buffered sampling).

    python examples/marl_env_groups.py --envs 4096 --steps 6000

A multi-agent step is two launches: the step kernel (one wave per env, waiting for memory about half of its life) and the
four-wave observation kernel.  Stepped as ONE batch they run one after the other.  `Engine.set_groups(2)` splits the same handle
into two halves with their own streams; `step_group(g, actions)` enqueues a half's two launches on its stream and returns, so the
step kernel of one half runs beside the observation kernel of the other (bench.py, rows c5_40x72 / c5_40x72_two_groups: 81 M ->
94 M env-steps/s at 4096 envs).  Envs do not interact, so the halves leave exactly the bytes the single batch leaves
(tests/test_parity_gpu.py::test_multi_agent_env_groups_step_like_one_batch); a learner that consumes group A's rows while group
B steps gets the overlap for free.  Work that reads a group's rows belongs on `engine.group_streams[g]`.

As eager launches the two halves need four kernel launches from the host inside ~43 us per iteration; a slow or busy host turns the
gain into a loss (one box of the round's measurements: 56 M with two groups against 92 M with one).  A rollout that keeps its
policy on the GPU captures `policy -> step_group` of each group in a HIP graph on the group's stream instead
(examples/fused_policy_rollout.py; bench.py --groups 2 --groups-graph 64): one host call per group and replay.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout without installing
from pgdrive_amd import MultiAgentRoundaboutVecEnv  # noqa: E402


def rollout(env, steps, groups, throttle):
    eng = env.engine
    N, A = env.num_envs, env.A
    dev = eng.obs.device
    ret = [torch.zeros(1, device=dev) for _ in range(max(groups, 1))]
    acts = [torch.rand((N, A, 2), device=dev) * 2 - 1 for _ in range(16)]  # open loop here; see fused_policy_rollout.py for a policy
    if throttle:
        for a in acts:
            a[..., 1] = a[..., 1] * 0.5 + 0.5  # mostly throttle: the roundabout fills up
    torch.cuda.synchronize()

    def iteration(k):
        # (the consumer below runs every 32nd step only: two torch ops and a stream switch per step cost the host more than the
        # GPU needs for the step -- a real consumer is a policy captured in a graph per group, see fused_policy_rollout.py)
        if groups <= 1:
            _, rew, _, _ = env.step(acts[k % 16])
            if k % 32 == 0:
                ret[0].add_(rew.sum())
            return
        for g in range(groups):
            _, rew, _, _ = eng.step_group(g, acts[(k + 5 * g) % 16])  # (the full action tensor: the group reads its own rows)
            if k % 32 == 0:
                with torch.cuda.stream(eng.group_streams[g]):  # the consumer of a group's rows runs on the group's stream
                    ret[g].add_(rew.sum())

    for k in range(3000):  # pre-roll: the roundabout fills up over the first thousands of steps
        iteration(k)
    for g in range(groups if groups > 1 else 0):
        eng.group_sync(g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        iteration(k)
    t_enq = time.perf_counter() - t0  # the host's share: every launch of the window enqueued
    for g in range(groups if groups > 1 else 0):
        eng.group_sync(g)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return N * steps / dt, float(sum(r.item() for r in ret)), t_enq / steps * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--throttle", action="store_true", help="mostly-throttle actions instead of uniform(-1, 1): more agents alive")
    args = ap.parse_args()
    out = {}
    for groups in (1, 2):
        env = MultiAgentRoundaboutVecEnv(dict(num_envs=args.envs, num_agents=40, seed=3))
        env.reset()
        if groups > 1:
            env.engine.set_groups(groups)
        out[groups] = rollout(env, args.steps, groups, args.throttle)
        status = env.slot_table()[0]
        print("agents alive per env %.1f" % float((status == 2).sum(axis=1).mean()), end="   ")
        print("%d group(s): %.1f M env-steps/s (%.0f M agent seats/s), host %.1f us per iteration, reward summed over every 32nd step %.1f   [%s]" %
              (groups, out[groups][0] / 1e6, out[groups][0] * env.A / 1e6, out[groups][2], out[groups][1], env.engine.describe_step()))
        env.close()


if __name__ == "__main__":
    main()
