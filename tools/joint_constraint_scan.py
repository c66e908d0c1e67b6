#!/usr/bin/env python3
"""The joint hypothesis round 4's study skipped: `crazyflie/*` and `evaluation/*` of the reference's training log are two
differently configured MDPs (they pay 0.289 and 1.299 reward per step) - so the nominal-Crazyflie evaluation may also START its
episodes from another distribution than the sampled-quadrotor one.  Is there a setting of the constants the reference tree does
not state - shared termination threshold, shared disturbance, and an initial tilt PER TAG - that reproduces share terminated,
episode length and time to failure of BOTH tags, with this repository's randomisation ranges untouched?

Oracle only (CPU; a study of the specification, not of the HIP path).  Stage 1 scans (termination_position x
disturbance_force_std x init_max_angle) on the sampled quadrotors against `evaluation/*`; stage 2 takes every (threshold,
disturbance) pair that has a match and scans the initial tilt on the nominal Crazyflie against `crazyflie/*`.

    python tools/joint_constraint_scan.py [--envs 16384] [--json profiles/r05_joint_constraints.json]

The answer decides nothing about l2f (only its sources can); it says whether the log CAN pin the constants: one surviving setting
would be a candidate default, several mean the log under-determines them, none means the structure assumed here is wrong."""
import argparse
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from oracle import oracle as O
    from raptor_amd.foundation_policy import load_weights
    from env_constraint_study import statistics
    w = load_weights()
    log = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_log.json")))
    tgt_cf, tgt_ev = log["pooled"]["crazyflie"]["last_100"], log["pooled"]["evaluation"]["last_100"]
    n = args.envs
    # tolerances: three standard errors of the log's pool plus three of this sample (share: sqrt(p q / n); time to failure: its
    # spread ~60 steps over sqrt(terminated episodes))
    def tol(t):
        p = t["share_terminated"]
        return (3 * t["share_terminated_se"] + 3 * (p * (1 - p) / n) ** 0.5,
                3 * t["terminated_episode_length_se"] + 3 * 60.0 / max(1.0, (p * n) ** 0.5))

    def fits(row, t):
        ts, tl = tol(t)
        lt = row["terminated_episode_length"]
        return abs(row["share_terminated"] - t["share_terminated"]) <= ts and lt is not None and abs(lt - t["terminated_episode_length"]) <= tl

    thresholds = [0.8, 1.0, 1.25, 1.5, 2.0]
    disturbances = [0.0, 0.08, 0.12, 0.16]
    angles = [1.5708, 1.66, 1.75, 1.83, 1.92, 2.0, 2.1]
    stage1, matches = [], []
    print(f"log evaluation/*: {tgt_ev['share_terminated']:.4f} terminated after {tgt_ev['terminated_episode_length']:.1f}   "
          f"log crazyflie/*: {tgt_cf['share_terminated']:.4f} after {tgt_cf['terminated_episode_length']:.1f}   ({n} envs per point)")
    for p, d, a in itertools.product(thresholds, disturbances, angles):
        over = dict(termination_position=p, disturbance_force_std=d, init_max_angle=a)
        ev = statistics(O, w, n, 1, 7, over)
        ok = fits(ev, tgt_ev)
        stage1.append({"change": over, "sampled_quadrotors": ev, "fits_evaluation_tags": ok})
        if ok:
            matches.append((p, d, a, ev))
            print(f"  evaluation fits: threshold {p} m, disturbance {d} m g, tilt {a:.2f} rad -> {ev['share_terminated']:.4f} after {ev['terminated_episode_length']}", flush=True)
    stage2, joint = [], []
    for p, d in sorted({(m[0], m[1]) for m in matches}):
        for a in angles:
            over = dict(termination_position=p, disturbance_force_std=d, init_max_angle=a)
            cf = statistics(O, w, n, 0, 3, over)
            ok = fits(cf, tgt_cf)
            stage2.append({"change": over, "nominal_crazyflie": cf, "fits_crazyflie_tags": ok})
            if ok:
                for m in matches:
                    if (m[0], m[1]) == (p, d):
                        joint.append({"termination_position": p, "disturbance_force_std": d, "init_max_angle_evaluation": m[2],
                                      "init_max_angle_crazyflie": a, "sampled_quadrotors": m[3], "nominal_crazyflie": cf})
                        print(f"  JOINT: threshold {p}, disturbance {d}, tilt evaluation {m[2]:.2f} / crazyflie {a:.2f}: "
                              f"crazyflie {cf['share_terminated']:.4f} after {cf['terminated_episode_length']}", flush=True)
    single = [j for j in joint if abs(j["init_max_angle_evaluation"] - j["init_max_angle_crazyflie"]) < 1e-6]
    verdict = ("no point of the grid reproduces even the evaluation/* tags alone (share AND time to failure): with this repository's "
               "randomisation ranges the family threshold x disturbance x tilt cannot be what separates the specification from the log"
               if not matches else "no setting of this family reproduces both tags" if not joint else
               f"{len(joint)} setting(s) reproduce both tags ({len(single)} with ONE initial tilt for both): "
               + ("the log pins a candidate" if len(joint) == 1 else "the log under-determines the constants"))
    print(verdict)
    if args.json:
        with open(args.json, "w") as fh:
            json.dump({"envs": n, "log_nominal_crazyflie": tgt_cf, "log_sampled_quadrotors": tgt_ev,
                       "grid": {"termination_position": thresholds, "disturbance_force_std": disturbances, "init_max_angle": angles},
                       "evaluation_matches": [{"termination_position": m[0], "disturbance_force_std": m[1], "init_max_angle": m[2], **m[3]} for m in matches],
                       "joint_matches": joint, "verdict": verdict,
                       # every stage-1 point, compactly: [threshold, disturbance, tilt, share, terminated-after]
                       "stage1": [[r["change"]["termination_position"], r["change"]["disturbance_force_std"], r["change"]["init_max_angle"],
                                   r["sampled_quadrotors"]["share_terminated"], r["sampled_quadrotors"]["terminated_episode_length"]] for r in stage1],
                       "stage1_closest": sorted(stage1, key=lambda r: abs(r["sampled_quadrotors"]["share_terminated"] - tgt_ev["share_terminated"]) / 0.004
                                                + abs((r["sampled_quadrotors"]["terminated_episode_length"] or 0) - tgt_ev["terminated_episode_length"]) / 8.0)[:5],
                       "stage2_points": len(stage2)}, fh, indent=1)
            fh.write("\n")


if __name__ == "__main__":
    main()
