#!/usr/bin/env python3
"""The core clock a fused-rollout launch runs at, against the idle time in front of it (rq_device_last_rollout_clock: the
median wave's shader-clock cycles over constant-rate ticks).  The chip drops its clock when it idles and takes ~100 us of
load to bring it back: what a short launch behind a synchronisation point pays beyond its own work.
    python tools/idle_clock.py [--steps 20] [--precision fp32]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raptor_amd.l2f as l2f                       # noqa: E402
from bench import Shard                            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--precision", default="fp32")
args = ap.parse_args()
device = l2f.Device()
sh = Shard(device, args.envs, 0)
sh.policy.set_precision(args.precision)
sh.rollout(3000, "fused")
device.set_rollout_timing(True)


def spin(us):
    t = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t:
        pass


print(f"{args.envs} envs, {args.precision}, launches of {args.steps} steps; median of {args.reps}")
for chain in (1, 2, 3, 5, 10):          # the LAST of `chain` launches enqueued back to back behind a 1 ms idle gap
    ms, ghz = [], []
    for _ in range(args.reps):
        device.synchronize()
        spin(1000)
        for _ in range(chain):
            sh.rollout(args.steps, "fused")
        ms.append(device.last_rollout_ms()); ghz.append(device.last_rollout_clock_ghz())
    print(f"launch {chain:2d} of a back-to-back chain behind 1 ms of idling: kernel {np.median(ms) * 1e3:7.2f} us, core clock {np.median(ghz):.3f} GHz")
for gap in (0, 5, 10, 20, 50, 100, 200, 500, 1000, 5000, 20000):
    ms, ghz = [], []
    for _ in range(args.reps):
        for _ in range(10):              # the chip at its loaded clock first
            sh.rollout(args.steps, "fused")
        device.synchronize()
        spin(gap)
        sh.rollout(args.steps, "fused")
        ms.append(device.last_rollout_ms()); ghz.append(device.last_rollout_clock_ghz())
    print(f"idle gap {gap:6d} us (+ synchronize and launch path) behind 10 launches: kernel {np.median(ms) * 1e3:7.2f} us, core clock {np.median(ghz):.3f} GHz")
