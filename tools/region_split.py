#!/usr/bin/env python3
"""Where the wall time of one timed region of `bench.py --steps 20` goes: the rollout call (host launch path), the
library's stream synchronize, torch.cuda.synchronize, against the kernel's own duration.
    python tools/region_split.py [--steps 20] [--reps 2000]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raptor_amd.l2f as l2f                       # noqa: E402
from bench import Shard                            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--reps", type=int, default=2000)
ap.add_argument("--spin", action="store_true", help="hipSetDeviceFlags(hipDeviceScheduleSpin) before anything runs")
ap.add_argument("--variants", default="lib+torch,torch only,lib only,query spin",
                help="comma-separated subset of the region shapes to time (tools/runtime_knobs.sh times 'torch only')")
args = ap.parse_args()
import ctypes                                      # noqa: E402
hip = ctypes.CDLL("libamdhip64.so")
if args.spin:
    torch.cuda.init()
    print("hipSetDeviceFlags(hipDeviceScheduleSpin) ->", hip.hipSetDeviceFlags(1))
device = l2f.Device()
sh = Shard(device, 65536, 0)
sh.rollout(5000, "fused")
device.synchronize()
pc = time.perf_counter


def med(xs):
    return float(np.median(xs)) * 1e6


stream = ctypes.c_void_p(int(device.stream))
for variant in [v.strip() for v in args.variants.split(",") if v.strip()]:
    call, s1, s2, tot = [], [], [], []
    for _ in range(args.reps):
        torch.cuda.synchronize()
        t0 = pc()
        sh.rollout(args.steps, "fused")
        t1 = pc()
        if variant == "query spin":
            while hip.hipStreamQuery(stream) != 0:
                pass
        elif variant != "torch only":
            device.synchronize()
        t2 = pc()
        if variant != "lib only":
            torch.cuda.synchronize()
        t3 = pc()
        call.append(t1 - t0); s1.append(t2 - t1); s2.append(t3 - t2); tot.append(t3 - t0)
    print(f"{variant:10s}: region {med(tot):7.2f} us = call {med(call):6.2f} + lib sync {med(s1):6.2f} + torch sync {med(s2):6.2f}")
device.set_rollout_timing(True)
ks = []
for _ in range(200):
    sh.rollout(args.steps, "fused")
    ks.append(device.last_rollout_ms() * 1e3)
device.set_rollout_timing(False)
print(f"kernel (own timestamps): {np.median(ks):.2f} us")
# an empty stream: what the two synchronizes cost with nothing to wait for
e1, e2 = [], []
for _ in range(args.reps):
    t0 = pc(); device.synchronize(); t1 = pc(); torch.cuda.synchronize(); t2 = pc()
    e1.append(t1 - t0); e2.append(t2 - t1)
print(f"idle: lib sync {med(e1):.2f} us, torch sync {med(e2):.2f} us")
