"""Closed loop with the policy network evaluated by the engine itself (pgd_mlp_policy): observation -> MLP -> action -> step, two
launches per iteration, optionally captured in a HIP graph.

    python examples/fused_policy_rollout.py --envs 4096 --steps 2000 [--weights policy.npz] [--graph]

The network is the shape of the reference's shipped PPO expert (pgdrive/examples/ppo_expert/numpy_expert.py: tanh MLP with two
256-wide hidden layers; weights as `kernel` arrays [in][out]).  `--weights` takes an .npz with the reference's key names
(default_policy/fc_1/kernel, .../bias, fc_2, fc_out) whose first layer has as many inputs as the observation has floats; without it
random weights are used (this is a throughput example).  Compare examples/graph_rollout.py, the same loop with torch ops as policy."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout without installing
from pgdrive_amd import PGDriveVecEnv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--weights", default=None)
    ap.add_argument("--graph", action="store_true", help="capture 4 iterations in one HIP graph and replay it")
    ap.add_argument("--bf16x3", action="store_true",
                    help="the split-bf16 kernel on prepared weights (pgd_mlp_prepare / pgd_mlp_policy_prepared): 16 bits of mantissa, "
                         "a third of the exact kernel's time")
    args = ap.parse_args()
    env = PGDriveVecEnv(dict(num_envs=args.envs, start_seed=1000, environment_num=100, auto_reset=True))
    eng, D = env.engine, env.obs_dim
    if args.weights:
        z = np.load(args.weights)
        w = [z["default_policy/fc_1/kernel"], z["default_policy/fc_1/bias"], z["default_policy/fc_2/kernel"], z["default_policy/fc_2/bias"],
             z["default_policy/fc_out/kernel"], z["default_policy/fc_out/bias"]]
        assert w[0].shape == (D, 256), "the first layer takes %d inputs, the observation has %d floats" % (w[0].shape[0], D)
    else:
        rng = np.random.default_rng(0)
        w = [rng.normal(0, D ** -0.5, (D, 256)), np.zeros(256), rng.normal(0, 1 / 16, (256, 256)), np.zeros(256),
             rng.normal(0, 1 / 16, (256, 2)), np.array([0.0, 0.5])]  # (a bias towards the throttle: the cars drive)
    weights = tuple(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda() for v in w)
    prepared = eng.mlp_prepare(weights) if args.bf16x3 else None  # (once per policy update)
    act = torch.zeros((args.envs, 1, 2), device="cuda")
    ret = torch.zeros(1, device="cuda")
    env.reset()

    def iteration():
        eng.mlp_policy(weights, act, final_tanh=True, prepared=prepared)  # reads the engine's observation buffer, writes the action buffer
        _, rew, _, _ = eng.step(act)
        ret.add_(rew.sum())

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(50):
            iteration()
        torch.cuda.synchronize()
        if args.graph:
            g = torch.cuda.CUDAGraph()
    if args.graph:
        with torch.no_grad(), torch.cuda.graph(g, stream=s):
            for _ in range(4):
                iteration()
        run, n = g.replay, args.steps // 4
        per = 4
    else:
        run, n, per = iteration, args.steps, 1
    with torch.cuda.stream(s), torch.no_grad():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("%d envs, fused policy%s: %.1f M env-steps/s (%.1f us per iteration); mean step reward %.4f" % (
        args.envs, " in a HIP graph" if args.graph else "", args.envs * n * per / dt / 1e6, dt / (n * per) * 1e6,
        float(ret) / (args.envs * (50 + (4 if args.graph else 0) + n * per))))
    env.close()


if __name__ == "__main__":
    main()
