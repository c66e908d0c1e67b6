#!/usr/bin/env python3
"""Soak: 100 000 auto-reset steps of 65 536 envs with observation noise and per-episode disturbances, fp32 and
bf16 actor; prints throughput and sanity figures (finite state, |q| = 1, share of terminated episodes).
    python tools/soak.py
"""
import os, sys, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raptor_amd.l2f as l2f
from raptor_amd.foundation_policy import Raptor
device = l2f.Device()
n = 65536
vector = l2f.vector(n)
rng, env, params, state = vector.VectorRng(), vector.VectorEnvironment(), vector.VectorParameters(), vector.VectorState()
vector.initialize_rng(device, rng, 123)
vector.initialize_environment(device, env)
cfg = env.config
cfg.noise_position = cfg.noise_orientation = 0.001
cfg.noise_linear_velocity = cfg.noise_angular_velocity = 0.002
cfg.disturbance_force_std, cfg.disturbance_torque_std = 0.05, 0.01
env.config = cfg
vector.sample_initial_parameters(device, env, params, rng)
vector.sample_initial_state(device, env, params, state, rng)
for prec in ("fp32", "bf16", "f16x2"):
    pol = Raptor(device, precision=prec); pol.reset()
    t0 = time.perf_counter()
    for k in range(100):
        vector.rollout(device, env, params, state, pol, rng, 1000, "fused", autoreset=True)
    device.synchronize()
    dt = time.perf_counter() - t0
    S = state.numpy(); H = pol.hidden_state(n)
    cnt, term, ret, ln = env.finished_counts(), env.finished_terminated(), env.finished_returns(), env.finished_lengths()
    print(f"[{prec}] 100k steps x {n} envs in {dt:.2f} s ({n*100000/dt:.3g} env-steps/s): finite state {np.isfinite(S).all()}, "
          f"hidden {np.isfinite(H).all()} |h|max {np.abs(H).max():.3f}, episodes/env {cnt.mean():.1f}, terminated share {term.sum()/cnt.sum():.4f}, "
          f"mean finished length {ln.mean():.1f}, mean finished return {ret.mean():.1f}, |q|-1 max {np.abs(np.linalg.norm(S[:,3:7],axis=1)-1).max():.2e}")
