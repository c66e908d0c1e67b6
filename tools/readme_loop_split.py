#!/usr/bin/env python3
"""Where the time of one README-loop iteration (README.md:96-99, N = 8, NumPy arrays across the boundary) goes: wall
time of each of the four calls, averaged; plus the same loop with the C entry points called directly (binding cost)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raptor_amd.l2f as l2f                       # noqa: E402
from raptor_amd.foundation_policy import Raptor    # noqa: E402
device = l2f.Device()
vector = l2f.vector(8)
rng, env = vector.VectorRng(), vector.VectorEnvironment()
params, state, next_state = vector.VectorParameters(), vector.VectorState(), vector.VectorState()
vector.initialize_rng(device, rng, 0); vector.initialize_environment(device, env)
vector.sample_initial_parameters(device, env, params, rng); vector.sample_initial_state(device, env, params, state, rng)
policy = Raptor(device); policy.reset()
obs = np.zeros((8, env.OBSERVATION_DIM), np.float32)
pc = time.perf_counter
T = np.zeros(5)
iters = 3000
for it in range(iters + 300):
    if it == 300:
        T[:] = 0
    t0 = pc(); vector.observe(device, env, params, state, obs, rng)
    t1 = pc(); x = obs[:, :22]
    t1b = pc(); action = policy.evaluate_step(x)
    t2 = pc(); vector.step(device, env, params, state, action, next_state, rng)
    t3 = pc(); state.assign(next_state)
    t4 = pc()
    T += (t1 - t0, t2 - t1b, t3 - t2, t4 - t3, t4 - t0)
T = T / iters * 1e6
print(f"observe {T[0]:.2f} us, evaluate_step {T[1]:.2f} us, step {T[2]:.2f} us, assign {T[3]:.2f} us, iteration {T[4]:.2f} us")
