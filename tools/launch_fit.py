#!/usr/bin/env python3
"""Fit T(n_steps) = a + b * n_steps for one fused-rollout launch (HIP events on the engine's stream).

a = per-launch cost (dispatch + weight/state prologue + epilogue), b = steady-state time per step.
    python tools/launch_fit.py [--envs 65536] [--precision fp32]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raptor_amd.l2f as l2f                       # noqa: E402
from bench import Shard                            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--precision", default="fp32")
ap.add_argument("--noise", action="store_true")
args = ap.parse_args()

device = l2f.Device()
sh = Shard(device, args.envs, 0, precision=args.precision)
if args.noise:
    cfg = sh.env.config
    cfg.noise_position, cfg.noise_orientation = 0.001, 0.001
    cfg.noise_linear_velocity, cfg.noise_angular_velocity = 0.002, 0.002
    sh.env.config = cfg
sh.rollout(2000, "fused")
device.synchronize()
ns = [1, 2, 5, 10, 20, 50, 100, 200, 500, 1000]
ts = []
for n in ns:
    best = 1e9
    for _ in range(7):
        device.timer_start()
        sh.rollout(n, "fused")
        best = min(best, device.timer_stop() * 1e3)
    ts.append(best)
    print(f"n_steps {n:5d}: {best:10.2f} us  ({best / n:8.3f} us/step)")
b, a = np.polyfit(ns, ts, 1)
print(f"fit: launch {a:.2f} us + {b:.4f} us/step   ({args.envs / b * 1e6:.4g} env-steps/s steady state)")
