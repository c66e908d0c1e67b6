#!/usr/bin/env python3
"""Soak of the resident executors (round 6): long README loops (README.md:96-99) and long runs of the policy alone (README.md:17-25),
with the executor and with the launches it replaces, compared bit for bit - every observation, action, state, hidden state, reward
(a NaN equals a NaN: the API loop does not reset an env that has diverged, its state goes non-finite after a few thousand steps, and
which NaN an instruction returns depends on its operand order - the one thing two compilations of the same source may differ in).

    python tools/resident_soak.py [--iters 20000] [--json out.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raptor_amd.l2f as l2f                                   # noqa: E402
from test_gpu_resident import _loop, _policy_loop       # noqa: E402


def same(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.dtype.kind != "f":
        return np.array_equal(a, b)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20000)
ap.add_argument("--json", default=None)
args = ap.parse_args()
device = l2f.Device(0)
out = {"iterations_per_case": args.iters, "loop": [], "policy": []}
bad = 0
for n in (1, 8, 12, 13, 64, 200, 256):
    t0 = time.perf_counter()
    off = _loop(device, n, args.iters, False, seed=n)
    on = _loop(device, n, args.iters, True, seed=n)
    ok = all(same(x, y) for x, y in zip(off[:6], on[:6]))
    bad += not ok
    out["loop"].append({"envs": n, "identical": bool(ok), **on[6], "seconds": round(time.perf_counter() - t0, 2)})
    print(out["loop"][-1], flush=True)
for b in (1, 2, 3, 8, 16):
    t0 = time.perf_counter()
    off = _policy_loop(device, b, args.iters, False, seed=b, wide=bool(b & 1))
    on = _policy_loop(device, b, args.iters, True, seed=b, wide=bool(b & 1))
    ok = same(off[0], on[0]) and same(off[1], on[1])
    bad += not ok
    out["policy"].append({"batch": b, "identical": bool(ok), **on[2], "seconds": round(time.perf_counter() - t0, 2)})
    print(out["policy"][-1], flush=True)
out["cases_that_differ"] = int(bad)
if args.json:
    with open(args.json, "w") as f:
        json.dump(out, f, indent=1)
sys.exit(1 if bad else 0)
