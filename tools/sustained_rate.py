#!/usr/bin/env python3
"""One line: sustained us per step of the fused rollout (median and min of 5 regions of `--launches` x 500-step launches back to
back) and the per-launch region of 20 steps, for same-box A/B runs alternating libraries (RAPTOR_QUAD_LIB):
    for r in 1 2 3; do for lib in "" scratch/variants/libraptor_quad_X.so; do RAPTOR_QUAD_LIB=$lib python tools/sustained_rate.py --envs 65536; done; done"""
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
ap.add_argument("--precision", default="fp32")
ap.add_argument("--launches", type=int, default=10)
args = ap.parse_args()
device = l2f.Device()
sh = Shard(device, args.envs, 0, precision=args.precision)
sh.rollout(2000, "fused")
per = []
for _ in range(5):
    device.synchronize()
    device.timer_start()
    for _ in range(args.launches):
        sh.rollout(500, "fused")
    per.append(device.timer_stop() * 1e3 / (500 * args.launches))
short = []
for _ in range(400):
    device.synchronize()
    t0 = time.perf_counter()
    sh.rollout(20, "fused")
    device.synchronize()
    short.append((time.perf_counter() - t0) * 1e6)
tag = os.path.basename(os.environ.get("RAPTOR_QUAD_LIB") or "product")
print(f"{tag:28s} {args.precision} {args.envs:8d} envs: sustained {np.median(per):.4f} us/step (min {min(per):.4f})   20-step region {np.median(short):.2f} us", flush=True)
