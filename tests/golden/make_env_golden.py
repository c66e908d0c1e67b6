#!/usr/bin/env python3
"""Spec-freeze fixture of the ENVIRONMENT half (this repository's own specification, DESIGN.md section 4 - NOT an l2f
vector: nothing under /root/reference can produce one).  It records what oracle/raptor_oracle.c
computes today for a small closed loop, so that a later change of the specification (operation order, a constant, the
RNG layout) cannot happen silently: tests/test_oracle_env.py::test_env_spec_fixture replays it on the CPU bit for bit,
tests/test_gpu_env.py::test_env_spec_fixture_on_the_gpu feeds its (state, action) pairs to the HIP kernels.

    python tests/golden/make_env_golden.py        # rewrites tests/golden/env_spec.npz; commit it with the change
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O                      # noqa: E402

N, STEPS, SEED, OFFSET = 16, 40, 20250419, 12345


def generate():
    w = np.fromfile(os.path.join(ROOT, "raptor_amd", "data", "raptor_policy.bin"), dtype="<f4")
    cfg = O.default_config()
    P = O.sample_initial_parameters(cfg, SEED, 0, OFFSET, N)
    st = O.Stats(N)
    S = O.sample_initial_state(cfg, SEED, st.episode, OFFSET, P)
    H = np.zeros((N, 16), np.float32)
    out = dict(params=P.copy(), state0=S.copy(), obs=[], act=[], state=[], reward=[], terminated=[])
    for k in range(STEPS):
        obs = O.observe(cfg, SEED, k, OFFSET, P, S)
        act = O.actor_batch_step(w, obs[:, :22], H)
        S, r, term = O.step(cfg, P, S, act)
        for key, v in (("obs", obs), ("act", act), ("state", S), ("reward", r), ("terminated", term)):
            out[key].append(np.array(v).copy())
    for key in ("obs", "act", "state", "reward", "terminated"):
        out[key] = np.stack(out[key])
    out["config_bytes"] = np.frombuffer(bytes(cfg), dtype=np.uint8).copy()
    out["meta"] = np.array([N, STEPS, SEED, OFFSET], np.int64)
    return out


if __name__ == "__main__":
    O.build()
    data = generate()
    path = os.path.join(ROOT, "tests", "golden", "env_spec.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, {k: v.shape for k, v in data.items()})
