"""Minimal use of the batched env: N PGDrive-v0 environments stepped with random actions on one MI355X.

    python examples/random_rollout.py --envs 32768 --steps 500

The engine step itself takes ~25 us for 4096 envs (bench.py: 166 M env-steps/s); in a Python loop like this one the
host-side launch cost of the surrounding torch ops (random actions, the `done` count) dominates at small N, so use a
large N per GPU -- the step kernel scales to 318 M env-steps/s at 262144 envs.
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
    ap.add_argument("--envs", type=int, default=32768)
    ap.add_argument("--steps", type=int, default=500)
    args = ap.parse_args()
    env = PGDriveVecEnv(dict(num_envs=args.envs, start_seed=1000, environment_num=100))  # PGDrive-v0: seeds 1000..1099, 1 ego + IDM traffic, 240 lidar beams
    obs = env.reset()  # cuda float32 [N, 274]
    episodes = torch.zeros(1, dtype=torch.int64, device=obs.device)  # counted on the device: no host sync inside the loop
    for _ in range(20):  # warm-up: the first calls of every torch op load their kernels
        env.step(torch.rand((args.envs, 2), device=obs.device) * 2 - 1)
        episodes += env.engine.done.view(-1).sum(0, keepdim=True) * 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        actions = torch.rand((args.envs, 2), device=obs.device) * 2 - 1  # any policy producing [N, 2] in [-1, 1]
        obs, reward, done, flags = env.step(actions)  # finished envs restart by themselves (PGD_F_RESET is set for them)
        episodes += done.sum(0, keepdim=True)  # (a 0-dim operand would be read back to the host as a scalar)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d env-steps in %.3f s = %.1f M env-steps/s, %d episodes finished" % (
        args.envs * args.steps, dt, args.envs * args.steps / dt / 1e6, int(episodes)))
    env.close()


if __name__ == "__main__":
    main()
