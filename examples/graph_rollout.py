"""Closed loop with a policy network, eager vs captured in a HIP graph.

    python examples/graph_rollout.py --envs 8192 --steps 500

With a policy in the loop a Python rollout is bound by the host: every torch op of the policy and the step itself cost a
launch from the interpreter (~100 us per iteration for a small MLP), more than the GPU needs for 8192 environments.  The
engine's step is one kernel launch on the caller's current stream with static output buffers, so `policy(obs) -> env.step`
can be captured once with `torch.cuda.graphs` and replayed: one host call per iteration (or per K iterations).
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout without installing
from pgdrive_amd import PGDriveVecEnv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--unroll", type=int, default=4, help="env steps captured per graph")
    args = ap.parse_args()
    torch.manual_seed(0)
    policy = torch.nn.Sequential(torch.nn.Linear(274, 256), torch.nn.Tanh(), torch.nn.Linear(256, 256), torch.nn.Tanh(),
                                 torch.nn.Linear(256, 2), torch.nn.Tanh()).cuda()
    env = PGDriveVecEnv(dict(num_envs=args.envs, start_seed=1000, environment_num=100))
    obs = env.reset()  # the engine's own observation buffer: env.step() rewrites it in place
    ret = torch.zeros(1, device=obs.device)

    def iteration():
        with torch.no_grad():
            act = policy(env.engine.obs.view(args.envs, -1))
            _, rew, done, _ = env.step(act)
            ret.add_(rew.sum())

    # eager
    for _ in range(30):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        iteration()
    torch.cuda.synchronize()
    eager = args.envs * args.steps / (time.perf_counter() - t0)

    # captured: warm up on a side stream (torch's capture recipe), then capture `unroll` iterations
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            iteration()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(args.unroll):
            iteration()
    torch.cuda.synchronize()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    n = max(1, args.steps // args.unroll)
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    graphed = args.envs * n * args.unroll / (time.perf_counter() - t0)
    print("%d envs, policy in the loop: eager %.1f M env-steps/s, HIP graph (%d steps per replay) %.1f M env-steps/s; "
          "mean step reward %.4f" % (args.envs, eager / 1e6, args.unroll, graphed / 1e6,
                                     float(ret) / (args.envs * (30 + args.steps + 3 + 10 * args.unroll + n * args.unroll + args.unroll))))
    env.close()


if __name__ == "__main__":
    main()
