#!/usr/bin/env python3
"""Kernel-only duration (the kernel's own begin/end timestamps, rq_device_set_rollout_timing) of fused rollout launches
of 1 ... 100 steps at 65 536 envs: what a launch costs before its first step, and per step once it runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raptor_amd.l2f as l2f
from bench import Shard
device = l2f.Device()
sh = Shard(device, 65536, 0)
sh.rollout(2000, "fused")
device.set_rollout_timing(True)
for n in (1, 2, 3, 5, 10, 20, 50, 100):
    ts = []
    for _ in range(30):
        sh.rollout(n, "fused")
        ts.append(device.last_rollout_ms() * 1e3)
    print(f"n_steps {n:4d}: kernel median {np.median(ts):8.2f} us  min {min(ts):8.2f}")
