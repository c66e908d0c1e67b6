#!/usr/bin/env python3
"""Run-to-run determinism of the fused rollout at scale: two identical batches, thousands of auto-reset steps in launches of
mixed length, every actor precision with and without the SampleAndSquash stage, every build that ships (fp32: one and two
waves per SIMD; bf16 / split-f16: one) - the final state, policy state and episode statistics must agree bit for bit (round 4
found a two-waves-per-SIMD bf16 build differing from run to run under another instruction scheduler; round 5 took it out of
the product, profiles/r05_bf16_two_wave_hunt.md; tests/test_gpu_fused.py::test_fused_rollout_is_deterministic is the short
version of this).
    python tools/determinism_soak.py [--steps 3000]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raptor_amd.l2f as l2f                       # noqa: E402
from bench import Shard                            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3000)
args = ap.parse_args()
device = l2f.Device()
bad = 0
for precision, sas in (("fp32", "off"), ("bf16", "off"), ("f16x2", "off"), ("fp32", "sample"), ("bf16", "sample"), ("bf16", "mean")):
    for n in (65536, 131072, 262144 + 129):
        a, b = Shard(device, n, 0, precision=precision), Shard(device, n, 0, precision=precision)
        if sas != "off":
            for sh in (a, b):
                sh.policy.set_sample_and_squash(sas, log_std_bias=np.full(4, -1.0, np.float32), seed=11)
        done, k = 0, 0
        while done < args.steps:
            c = (1, 7, 20, 500, 133)[k % 5]
            a.rollout(c, "fused"); b.rollout(c, "fused")
            done += c; k += 1
        rows = int((a.state.numpy() != b.state.numpy()).any(axis=1).sum())
        rows += int((a.policy.hidden_state(n) != b.policy.hidden_state(n)).any(axis=1).sum())
        rows += int((a.env.finished_returns() != b.env.finished_returns()).sum() + (a.env.finished_counts() != b.env.finished_counts()).sum())
        bad += rows
        print(f"{precision:6s} sas {sas:6s} {n:7d} envs x {done} steps in {k} launches: {rows} differing rows", flush=True)
print("deterministic" if bad == 0 else f"NOT deterministic: {bad}")
sys.exit(0 if bad == 0 else 1)
