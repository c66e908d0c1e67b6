#!/usr/bin/env python3
"""Throughput of the teacher-bank relabel (rq_trajectory_relabel_teachers): n_envs x T recorded steps labelled by
n_teachers MLP teachers (22-64-64-4) in one launch: exact-f32, bf16 and split-f16 MFMA paths.

    python tools/teacher_rate.py [--envs 65536] [--steps 500] [--teachers 1000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raptor_amd.l2f as l2f                       # noqa: E402
from bench import Shard                            # noqa: E402
from raptor_amd.teachers import TeacherBank, balanced_teacher_assignment, parameter_count   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--teachers", type=int, default=1000)
ap.add_argument("--hidden", type=int, default=64)
ap.add_argument("--assignment", choices=("balanced", "contiguous"), default="balanced",
                help="balanced: raptor_amd.teachers.balanced_teacher_assignment (whole 64-env tiles per teacher where the "
                     "counts allow); contiguous: env e -> teacher e*T//n (65.536 envs per teacher at T=1000: ragged tiles)")
ap.add_argument("--only-layers", action="store_true", help="the dense-stack kernel only (tools/teacher_traffic.sh profiles this)")
args = ap.parse_args()

device = l2f.Device()
sh = Shard(device, args.envs, 0)
tr = sh.vector.Trajectory(sh.env, args.steps)
sh.vector.rollout(device, sh.env, sh.params, sh.state, sh.policy, sh.rng, args.steps, "fused", autoreset=True, trajectory=tr)
rng = np.random.default_rng(0)
H = args.hidden
W = (rng.standard_normal((args.teachers, parameter_count(22, H, H))) * 0.1).astype(np.float32)
if args.assignment == "balanced":
    ids = balanced_teacher_assignment(args.envs, args.teachers)
else:
    ids = (np.arange(args.envs) * args.teachers // args.envs).astype(np.uint32)
flop = 2 * (22 * H + H * H + H * 4) * args.envs * args.steps
out = {"envs": args.envs, "steps": args.steps, "teachers": args.teachers, "assignment": args.assignment, "topology": f"22-{H}-{H}-4",
       "flop_per_label": 2 * (22 * H + H * H + H * 4)}
for prec, peak in (() if args.only_layers else (("fp32", 157.3), ("bf16", 2500.0), ("f16x2", 2500.0))):    # f16x2: useful FLOP; 3x as many are issued
    bank = TeacherBank(device, W, 22, H, H, "relu", "identity", precision=prec)
    tr.relabel_teachers(bank, ids, fetch=False)
    device.synchronize()
    best, best_dev = 1e9, 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        device.timer_start()
        tr.relabel_teachers(bank, ids, fetch=False)
        best_dev = min(best_dev, device.timer_stop() * 1e-3)      # stream time: tile upload + kernel
        best = min(best, time.perf_counter() - t0)                 # + grouping the envs by teacher on the host
    wall_ms, best = best * 1e3, best_dev
    out[prec] = {"ms": round(best * 1e3, 3), "wall_ms_incl_host_grouping": round(wall_ms, 3), "labels_per_s": round(args.envs * args.steps / best, 1),
                 "TFLOPs": round(flop / best / 1e12, 2), "peak_TFLOPs": peak, "frac_of_mfma_peak": round(flop / best / 1e12 / peak, 4),
                 "obs_GBps": round(args.envs * args.steps * (88 + 16) / best / 1e9, 1)}
# the dense-stack kernel for teachers outside the register-stationary family (round 5; LDS-resident since round 6): fp32 only
from raptor_amd.teachers import layers_parameter_count          # noqa: E402
for widths in ([128, 128, 128], [128, 128], [64, 64, 64], [128]):
    Wl = (rng.standard_normal((args.teachers, layers_parameter_count(22, widths))) * 0.05).astype(np.float32)
    bank = TeacherBank.from_layers(device, Wl, 22, widths, "relu", "identity")
    dims = [22] + widths + [4]
    fl = 2 * sum(dims[i + 1] * dims[i] for i in range(len(dims) - 1))
    for _ in range(2):
        tr.relabel_teachers(bank, ids, fetch=False)
    device.synchronize()
    best = 1e9
    for _ in range(4):
        device.timer_start()
        tr.relabel_teachers(bank, ids, fetch=False)
        best = min(best, device.timer_stop() * 1e-3)
    tf = fl * args.envs * args.steps / best / 1e12
    out["layers_22-" + "-".join(map(str, widths)) + "-4"] = {"ms": round(best * 1e3, 3), "labels_per_s": round(args.envs * args.steps / best, 1),
                                                            "flop_per_label": fl, "TFLOPs": round(tf, 2), "frac_of_f32_mfma_peak": round(tf / 157.3, 4)
                                                            }
    del bank
print(json.dumps(out))
