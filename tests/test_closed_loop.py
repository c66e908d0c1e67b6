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


# Closed-loop statistics of the shipped policy in the REAL l2f, last record of the reference's training log
# (`logs.tfevents` inside /root/reference/data/raptor-policy-checkpoint.tar.gz, tags evaluation/* on sampled
# quadrotors): share of episodes ended by termination and mean episode length of 500.
REFERENCE_LOG = {"share_terminated": 0.042, "episode_length": 482.8,
                 # also in the log, not asserted: last-20-epoch means 0.0417 / 483.3,
                 # episode_length/std 65.1, return 619.0 +- 112.8 (reward constants unknown here)
                 "episode_length_std": 65.1, "return_mean": 619.0, "return_std": 112.8}


def test_closed_loop_statistics_against_the_reference_training_log(oracle, weights):
    """The one MDP constant this comparison needs and the tree does not state is the position termination
    threshold; with 1 m this simulator + the shipped policy reproduce both logged statistics (measured on the
    MI355X with 65 536 quadrotors: 0.0409 and 484.2; with the default 3 m: 0.016 / 495.4, with 0.6 m: 0.18 / 416).
    A statistical, one-parameter cross-check - not a parity claim - but it ties the restated dynamics,
    parameter distribution and initial-state distribution to numbers produced by the reference itself."""
    O = oracle
    n = 16384
    cfg = O.default_config()
    cfg.termination_position = 1.0
    P = O.sample_initial_parameters(cfg, 7, 0, 0, n)
    st = O.Stats(n)
    S = O.sample_initial_state(cfg, 7, st.episode, 0, P)
    H = np.zeros((n, 16), np.float32)
    O.rollout(cfg, weights, 7, 0, 0, P, S, H, 500, 0, st, O.max_threads())
    assert (st.fin_counts == 1).all()
    share, length = st.fin_terminated.mean(), st.fin_lengths.mean()
    assert abs(share - REFERENCE_LOG["share_terminated"]) < 0.012, share
    assert abs(length - REFERENCE_LOG["episode_length"]) < 5.0, length
