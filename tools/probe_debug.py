#!/usr/bin/env python3
"""Kernel-level span of 20-step fused launches (65 536 envs) against what the GPU did just before: fresh process, after torch
came up, right after 2 000 synchronised regions, after half a second of idling, after 5 000 busy steps.  The clock state
follows the recent load with a time constant of milliseconds: 70.7 us ... 78.8 us for the same launch (round 3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raptor_amd.l2f as l2f
from bench import Shard
device = l2f.Device()
sh = Shard(device, 65536, 0)
sh.rollout(2000, "fused")
def loop(tag, n=50, pre=None):
    device.set_rollout_timing(True)
    ts = []
    for _ in range(n):
        if pre: pre()
        sh.rollout(20, "fused")
        ts.append(device.last_rollout_ms() * 1e3)
    device.set_rollout_timing(False)
    print(f"{tag:40s} mean {np.mean(ts):6.2f} median {np.median(ts):6.2f} min {min(ts):6.2f} max {max(ts):6.2f}", flush=True)
loop("A fresh process, no torch")
loop("A again")
import torch
torch.cuda.set_device(0); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
loop("B torch initialised")
def regions(k):
    for _ in range(k):
        device.synchronize(); torch.cuda.synchronize()
        sh.rollout(20, "fused")
        device.synchronize(); torch.cuda.synchronize()
regions(2000)
loop("C after 2000 synced regions")
loop("C again")
loop("D with device+torch sync before each", pre=lambda: (device.synchronize(), torch.cuda.synchronize()))
time.sleep(0.5)
loop("E after 0.5 s idle")
sh.rollout(5000, "fused"); device.synchronize()
loop("F after 5000 busy steps")
