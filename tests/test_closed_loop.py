"""Functional pin of the env conventions: the SHIPPED policy — trained in the
real l2f — must stabilise the restated simulator, and must fail when a convention the
reference states (README.md:23-27) is violated."""
import numpy as np
import pytest


def _closed_loop(O, weights, n=256, dr=1, obs_hook=None, act_hook=None, steps=500, seed=0):
    cfg = O.default_config()
    cfg.domain_randomization = dr
    cfg.termination_enabled = 0
    P = O.sample_initial_parameters(cfg, seed, 0, 0, n)
    ep = np.zeros(n, np.uint32)
    S = O.sample_initial_state(cfg, seed, ep, 0, P)
    H = np.zeros((n, 16), np.float32)
    alive = np.ones(n, bool)
    for t in range(steps):
        obs = O.observe(cfg, seed, t, 0, P, S)
        if obs_hook is not None:
            obs = obs_hook(obs)
        act = O.actor_batch_step(weights, obs, H)
        if act_hook is not None:
            act = act_hook(act)
        S, _, _ = O.step(cfg, P, S, act)
        with np.errstate(invalid="ignore"):
            alive &= np.isfinite(S[:, :3]).all(axis=1) & (np.abs(np.nan_to_num(S[:, :3], nan=1e9)).max(axis=1) < 3.0)
    return alive.mean(), S, P


def test_policy_hovers_nominal_crazyflie(oracle, weights):
    alive, S, P = _closed_loop(oracle, weights, n=128, dr=0)
    assert alive >= 0.98
    assert np.median(np.linalg.norm(S[:, :3], axis=1)) < 0.05
    # settles at the analytic hover command
    assert np.allclose(S[:, 17:21].mean(), P[0, 25], atol=0.01)


def test_policy_stabilises_randomised_quadrotors(oracle, weights):
    alive, S, _ = _closed_loop(oracle, weights, n=512, dr=1)
    assert alive >= 0.93          # 0.957 measured on this distribution
    assert np.median(np.linalg.norm(S[:, :3], axis=1)) < 0.1


@pytest.mark.parametrize("violation", ["transposed_R", "rotated_motors", "mirrored_motors"])
def test_policy_fails_when_convention_is_violated(oracle, weights, violation):
    def transposed(obs):
        o = obs.copy()
        o[:, 3:12] = obs[:, 3:12].reshape(-1, 3, 3).transpose(0, 2, 1).reshape(-1, 9)
        return o
    hooks = {
        "transposed_R": dict(obs_hook=transposed),
        "rotated_motors": dict(act_hook=lambda a: np.ascontiguousarray(np.roll(a, 1, axis=1))),
        "mirrored_motors": dict(act_hook=lambda a: np.ascontiguousarray(a[:, [3, 2, 1, 0]])),
    }[violation]
    alive, _, _ = _closed_loop(oracle, weights, n=64, dr=0, **hooks)
    assert alive < 0.2
