#!/usr/bin/env python3
"""Throughput of the fused rollout WITH trajectory recording: every step stores
the 22 policy inputs, 4 actions, reward and done code of every env = 109 B per env-step.
    python tools/record_rate.py [--envs 65536] [--steps 100] [--precision fp32]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raptor_amd.l2f as l2f                       # noqa: E402
from bench import Shard                            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--precision", default="fp32")
args = ap.parse_args()
device = l2f.Device()
sh = Shard(device, args.envs, 0, precision=args.precision)
traj = sh.vector.Trajectory(sh.env, args.steps)
v = sh.vector
for _ in range(10):
    sh.rollout(500, "fused")                       # clocks
best = 1e9
for _ in range(5):
    traj.reset()
    device.timer_start()
    v.rollout(device, sh.env, sh.params, sh.state, sh.policy, sh.rng, args.steps, "fused", autoreset=True,
              trajectory=traj)
    best = min(best, device.timer_stop())
bytes_per = 4 * (22 + 4 + 1) + 1
rate = args.envs * args.steps / (best * 1e-3)
print(f"{args.envs} envs x {args.steps} steps recorded: {best:.3f} ms -> {rate:.4g} env-steps/s, "
      f"{rate * bytes_per / 1e9:.0f} GB/s of trajectory stores ({best * 1e3 / args.steps:.2f} us/step)")
