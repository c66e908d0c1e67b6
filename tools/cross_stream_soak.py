#!/usr/bin/env python3
"""Do the fp32 kernels give the same results whatever ANOTHER stream of the same GPU is running?

Round 5 (profiles/r05_bf16_two_wave_hunt.md): gfx950 misreads one operand of a packed-fp32 instruction of one op_sel form in lanes
48..63 while another wave of the same SIMD executes a 16-bit MFMA.  Inside this library only the (retired) two-waves-per-SIMD bf16
build could meet that with its own waves - but the fp32 kernels held the form too, and a wave of theirs fits beside a wave of the bf16
fused rollout (302 - 389 registers) of another rq_device on the same GPU.  This tool makes that happen: device A rolls bf16 episodes
out without pause on its stream, device B repeats one fp32 workload of API-granular kernels (observe -> evaluate_step -> step under a
hipGraph, `chained` mode) from the same start and compares every repetition bit for bit with what it gives while A is idle.

    python tools/cross_stream_soak.py [--reps 60]                      the product library: must print 0
    RAPTOR_QUAD_LIB=scratch/variants/libraptor_quad_NP.so python tools/cross_stream_soak.py       (NP: built with RQ_NO_OPSEL_REWRITE=1)
"""
import argparse
import collections
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raptor_amd.l2f as l2f                       # noqa: E402
from bench import Shard                            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--steps", type=int, default=200, help="steps of the fp32 workload per repetition")
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--victim", default="fp32", help="precision of the workload that is checked (fp32 | bf16 | f16x2)")
ap.add_argument("--mode", default="chained", help="chained (API-granular kernels under a hipGraph) | fused")
ap.add_argument("--aggressor", default="bf16", help="precision of the other stream's fused rollouts (bf16 | f16x2 | fp32 | none)")
args = ap.parse_args()
tag = os.path.basename(os.environ.get("RAPTOR_QUAD_LIB", "product"))

dev_a, dev_b = l2f.Device(0), l2f.Device(0)        # two engines on one GPU: a stream each


def workload():
    sh = Shard(dev_b, args.envs, 0, seed=7, precision=args.victim)
    sh.rollout(args.steps, args.mode)
    return np.concatenate([sh.state.numpy(), sh.policy.hidden_state(args.envs)], axis=1)


quiet = workload()
assert (quiet.view(np.uint32) == workload().view(np.uint32)).all(), "the workload is not deterministic on an idle GPU"

stop = threading.Event()
launched = [0]


failed = []


def aggressor():
    try:
        sh = Shard(dev_a, args.envs, 0, seed=3, precision=args.aggressor)
        while not stop.is_set():
            sh.rollout(2000, "fused")          # a few milliseconds per launch, auto-reset: the stream is busy almost all the time
            dev_a.synchronize()
            launched[0] += 1
    except Exception as e:                     # noqa: BLE001  (reported by the main thread)
        failed.append(repr(e))


t = None
if args.aggressor != "none":
    t = threading.Thread(target=aggressor, daemon=True)
    t.start()
    import time
    deadline = time.monotonic() + 120.0
    while launched[0] < 3 and not failed and time.monotonic() < deadline:
        time.sleep(0.001)
    if failed or launched[0] < 3:
        stop.set()
        sys.exit("the other stream's rollouts did not start: " + (failed[0] if failed else "timeout"))
bad_envs, bad_reps, quarters = 0, 0, collections.Counter()
for rep in range(args.reps):
    got = workload()
    d = (got.view(np.uint32) != quiet.view(np.uint32)).any(axis=1)
    if d.any():
        bad_reps += 1
        bad_envs += int(d.sum())
        quarters.update((np.nonzero(d)[0] % 64 // 16).tolist())
stop.set()
if t is not None:
    t.join(60.0)
if failed:
    sys.exit("the other stream's rollouts failed: " + failed[0])
print(f"[{tag}] {args.reps} repetitions of {args.steps} {args.victim} {args.mode} steps on {args.envs} envs beside {launched[0]} {args.aggressor} rollouts on another stream: "
      f"{bad_reps} repetitions differ from the idle-GPU result, {bad_envs} envs; by quarter of the wave {dict(sorted(quarters.items()))}")
sys.exit(1 if bad_envs else 0)
