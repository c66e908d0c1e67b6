#!/usr/bin/env python3
"""Which of the MDP constants the reference tree does not state are constrained by the numbers its training log holds
(tests/golden/reference_log.json, made by tests/golden/make_reference_log.py from logs.tfevents)?

    python tools/env_constraint_study.py [--envs 32768] [--json profiles/r04_env_constraints.json]

For the specification's default and for single-constant changes, the shipped policy is rolled out in the ORACLE (CPU; this
is a study of the specification, not of the HIP path) on the nominal Crazyflie (the log's crazyflie/* tags) and on the
domain-randomised quadrotors (evaluation/*), one 500-step episode per env, and three statistics are compared with the
log's pooled late-epoch values: share of episodes ended by termination, mean episode length, and the mean length of
the TERMINATED episodes ((L - (1 - s) 500) / s).  The third one is what round 4 adds: it separates "more failures of the
same kind" (harder initial tilt, a disturbance) from "failures that end sooner" (a tighter threshold)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CANDIDATES = [
    ("specification default", {}),
    ("init_max_angle 1.75 rad (100 deg)", dict(init_max_angle=1.75)),
    ("init_max_angle 1.83 rad (105 deg)", dict(init_max_angle=1.83)),
    ("init_max_angle 1.9 rad (109 deg)", dict(init_max_angle=1.9)),
    ("disturbance_force_std 0.12 m g", dict(disturbance_force_std=0.12)),
    ("disturbance_force_std 0.16 m g", dict(disturbance_force_std=0.16)),
    ("disturbance_force_std 0.19 m g", dict(disturbance_force_std=0.19)),
    ("termination_position 0.75 m", dict(termination_position=0.75)),
    ("termination_position 0.8 m", dict(termination_position=0.8)),
    ("init_max_position 0.8 m", dict(init_max_position=0.8)),
    ("init_max_linear_velocity 1.5 m/s", dict(init_max_linear_velocity=1.5)),
    ("termination_angular_velocity 10 rad/s", dict(termination_angular_velocity=10.0)),
    ("termination_linear_velocity 2 m/s", dict(termination_linear_velocity=2.0)),
    ("init_guidance 0", dict(init_guidance=0.0)),
]


def statistics(O, weights, n, dr, seed, over):
    cfg = O.default_config()
    cfg.domain_randomization = dr
    for k, v in over.items():
        setattr(cfg, k, v)
    P = O.sample_initial_parameters(cfg, seed, 0, 0, n)
    st = O.Stats(n)
    S = O.sample_initial_state(cfg, seed, st.episode, 0, P)
    H = np.zeros((n, 16), np.float32)
    O.rollout(cfg, weights, seed, 0, 0, P, S, H, 500, 0, st, O.max_threads())
    term = st.fin_terminated.astype(bool)
    L = st.fin_lengths.astype(np.float64)
    share = term.mean()
    return {"share_terminated": round(float(share), 4), "episode_length": round(float(L.mean()), 1),
            "terminated_episode_length": round(float(L[term].mean()), 1) if term.any() else None,
            "episode_length_std": round(float(L.std()), 1),
            "reward_per_step": round(float((st.fin_returns[~term] / L[~term]).mean()), 3) if (~term).any() else None}


def verdict(row, target, tol_share, tol_lt):
    ok_s = abs(row["share_terminated"] - target["share_terminated"]) <= tol_share
    lt = row["terminated_episode_length"]
    ok_l = lt is not None and abs(lt - target["terminated_episode_length"]) <= tol_lt
    return ("share " + ("ok" if ok_s else "off")) + ", terminated-after " + ("ok" if ok_l else "off")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=32768)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from oracle import oracle as O
    from raptor_amd.foundation_policy import load_weights
    w = load_weights()
    log = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_log.json")))
    tgt_cf, tgt_ev = log["pooled"]["crazyflie"]["last_100"], log["pooled"]["evaluation"]["last_100"]
    print("log, nominal Crazyflie  (last 100 epochs x 100 episodes):", {k: tgt_cf[k] for k in ("share_terminated", "episode_length", "terminated_episode_length")})
    print("log, sampled quadrotors (last 100 epochs x 10 000 episodes):", {k: tgt_ev[k] for k in ("share_terminated", "episode_length", "terminated_episode_length")})
    rows = []
    for name, over in CANDIDATES:
        cf = statistics(O, w, args.envs, 0, 3, over)
        ev = statistics(O, w, args.envs, 1, 7, over)
        # tolerances: three standard errors of the log's pool plus this sample's own (share: sqrt(p q / n))
        v_cf = verdict(cf, tgt_cf, 3 * tgt_cf["share_terminated_se"] + 0.003, 3 * tgt_cf["terminated_episode_length_se"] + 3.0)
        v_ev = verdict(ev, tgt_ev, 0.004, 6.0)
        rows.append({"candidate": name, "change": over, "nominal_crazyflie": cf, "sampled_quadrotors": ev,
                     "against_crazyflie_tags": v_cf, "against_evaluation_tags": v_ev})
        print(f"{name:40s} | CF  {cf['share_terminated']:.4f} {cf['episode_length']:6.1f} {cf['terminated_episode_length']}  [{v_cf}]"
              f" | DR  {ev['share_terminated']:.4f} {ev['episode_length']:6.1f} {ev['terminated_episode_length']}  [{v_ev}]"
              f" | r/step {cf['reward_per_step']} / {ev['reward_per_step']}")
    if args.json:
        with open(args.json, "w") as fh:
            json.dump({"envs": args.envs, "log_nominal_crazyflie": tgt_cf, "log_sampled_quadrotors": tgt_ev,
                       "log_reward_per_step": {k: v["reward_per_step"] for k, v in log["return_regression_last_500"].items()},
                       "rows": rows}, fh, indent=1)
            fh.write("\n")


if __name__ == "__main__":
    main()
