#!/usr/bin/env python3
"""Throughput of Raptor.evaluate_sequence (one launch for a whole [T, B, 22] observation tensor on the device).
    python tools/sequence_rate.py [--batch 65536] [--steps 200] [--precision fp32]
Algorithmic bytes per (step, batch element): 88 read + 16 written = 104; 3 904 FLOP on the matrix cores.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raptor_amd.l2f as l2f                       # noqa: E402
from raptor_amd.foundation_policy import Raptor    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--precision", default="fp32")
args = ap.parse_args()
device = l2f.Device()
p = Raptor(device, precision=args.precision)
p.reset()
x = torch.randn(args.steps, args.batch, 22, device="cuda:0")
for _ in range(5):
    p.evaluate_sequence(x)
best = 1e9
for _ in range(5):
    device.timer_start()
    p.evaluate_sequence(x)
    best = min(best, device.timer_stop())
rate = args.batch * args.steps / (best * 1e-3)
print(f"[{args.precision}] {args.steps} x {args.batch}: {best:.3f} ms -> {rate:.4g} policy steps/s, "
      f"{rate * 104 / 1e9:.0f} GB/s algorithmic ({rate * 104 / 8e12:.2f} of 8 TB/s), "
      f"{rate * 3904 / 1e12:.1f} TFLOP/s on the matrix cores ({best * 1e3 / args.steps:.2f} us/step)")
