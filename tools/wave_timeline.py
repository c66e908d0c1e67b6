#!/usr/bin/env python3
"""Where the time of one fused-rollout launch goes, wave by wave: every wave of a timed launch leaves the tick at which it
came in and went out (rq_device_last_rollout_waves).  Per die: how long the launch takes to get all its waves running
(spread of the arrival ticks), how long one wave runs, how far apart the waves finish.
    python tools/wave_timeline.py [--envs 65536] [--steps 1 2 5 20]
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
ap.add_argument("--steps", type=int, nargs="+", default=[1, 2, 5, 10, 20, 50, 200, 500])
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--precision", default="fp32")
ap.add_argument("--gap-ms", type=float, default=0.0, help="idle time in front of every launch")
args = ap.parse_args()
device = l2f.Device()
sh = Shard(device, args.envs, 0)
sh.policy.set_precision(args.precision)
sh.rollout(3000, "fused")
device.set_rollout_timing(True)
US = 1e-2            # 100 MHz ticks -> us
for n_steps in args.steps:
    rows, clocks = [], []
    for _ in range(args.reps):
        if args.gap_ms > 0:
            device.synchronize()
            time.sleep(args.gap_ms * 1e-3)
        sh.rollout(n_steps, "fused")
        t_in, t_out, xcd, t_l0, t_l1 = device.last_rollout_waves()
        clocks.append(device.last_rollout_clock_ghz())
        per = []
        for x in range(8):
            m = xcd == x
            if not m.any():
                continue
            a, b = t_in[m].astype(np.int64), t_out[m].astype(np.int64)
            l0, l1 = t_l0[m].astype(np.int64), t_l1[m].astype(np.int64)
            t0 = a.min()
            per.append((a.max() - t0, np.median(b - a), (b - a).min(), (b - a).max(), b.max() - b.min(), b.max() - t0,
                        np.median(a - t0), int(m.sum()), np.median(l0 - a), np.median(l1 - l0), np.median(b - l1),
                        (l1 - l0).max() - np.median(l1 - l0)))
        rows.append(np.mean(per, axis=0))
    r = np.median(np.array(rows), axis=0)
    print(f"n_steps {n_steps:3d}: waves/die {r[7]:.0f} | arrival spread {r[0] * US:6.2f} us (median wave arrives {r[6] * US:5.2f} us after the first)"
          f" | one wave runs {r[1] * US:6.2f} us (min {r[2] * US:6.2f}, max {r[3] * US:6.2f}) | finish spread {r[4] * US:6.2f} us"
          f" | first in -> last out {r[5] * US:6.2f} us\n"
          f"             median wave: prologue {r[8] * US:5.2f} us, steps {r[9] * US:6.2f} us, epilogue {r[10] * US:5.2f} us;"
          f" slowest wave's steps take {r[11] * US:5.2f} us longer than the median's\n"
          f"             core clock over the median wave's steps {np.median(clocks):.3f} GHz"
          f" = {r[9] * US * np.median(clocks) * 1e3 / n_steps:.0f} cycles per step")
