#!/usr/bin/env python3
"""The README loop (README.md:96-99, NumPy arrays at every call) with and without the resident executor of the small-batch loop
(rq_device_set_resident): the same bits in every observation, action and state, what the executor did, and the time per iteration.

    python tools/resident_check.py [--envs 8] [--iters 3000]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raptor_amd.l2f as l2f                       # noqa: E402
from raptor_amd.foundation_policy import Raptor    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=8)
ap.add_argument("--iters", type=int, default=3000)
ap.add_argument("--check", type=int, default=300, help="iterations whose observations / actions / states are compared bit for bit")
args = ap.parse_args()


def run(resident, iters, record):
    device = l2f.Device()
    device.set_resident(resident)
    vector = l2f.vector(args.envs)
    rng, env = vector.VectorRng(), vector.VectorEnvironment()
    params, state, next_state = vector.VectorParameters(), vector.VectorState(), vector.VectorState()
    vector.initialize_rng(device, rng, 0)
    vector.initialize_environment(device, env)
    vector.sample_initial_parameters(device, env, params, rng)
    vector.sample_initial_state(device, env, params, state, rng)
    policy = Raptor(device)
    policy.reset()
    obs = np.zeros((args.envs, env.OBSERVATION_DIM), np.float32)
    log = []
    t0 = None
    for it in range(iters + 200):
        if it == 200:
            device.synchronize() if not resident else None
            t0 = time.perf_counter()
        vector.observe(device, env, params, state, obs, rng)
        action = policy.evaluate_step(obs[:, :22])
        vector.step(device, env, params, state, action, next_state, rng)
        state.assign(next_state)
        if record and it < args.check:
            log.append((obs.copy(), action.copy()))
    us = (time.perf_counter() - t0) / iters * 1e6
    stats = device.resident()
    if resident:
        stats["last_command_us"] = device.resident_timing_us()
    final = state.numpy().copy()
    hidden = policy.hidden_state(args.envs).copy()
    rewards = env.rewards().copy()
    return us, stats, log, final, hidden, rewards


us_off, st_off, log_off, fin_off, hid_off, rew_off = run(False, args.iters, True)
us_on, st_on, log_on, fin_on, hid_on, rew_on = run(True, args.iters, True)
bad = sum(int(not (np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))))
          for a, b in zip(log_off, log_on))
same_end = (np.array_equal(fin_off.view(np.uint32), fin_on.view(np.uint32)) and np.array_equal(hid_off.view(np.uint32), hid_on.view(np.uint32))
            and np.array_equal(rew_off.view(np.uint32), rew_on.view(np.uint32)))
print(f"{args.envs} envs, {args.iters} iterations: launches {us_off:.2f} us per iteration, resident executor {us_on:.2f} us per iteration")
print(f"resident executor: {st_on}; without: {st_off}")
print(f"iterations whose observation or action differ: {bad} of {len(log_off)}; final state, hidden state and rewards identical: {same_end}")
sys.exit(0 if bad == 0 and same_end and st_on['commands'] > 0 else 1)
