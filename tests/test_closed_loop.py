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


# The same log holds a second record, parameter-free on the dynamics side: the shipped policy on the NOMINAL Crazyflie
# (tags crazyflie/*, last record; last-20-epoch means 0.0425 / 481.1).  The terminated episodes' mean length follows
# from the two: (477.4 - 0.95 * 500) / 0.05 = 48 steps.  crazyflie/return/* are not comparable: that environment carries a
# termination penalty of about -100 (first epoch: return -102 +- 5.8 at length 29.7 +- 8.3 with every episode terminated -
# the spread of the return is far below length spread x reward, so a constant dominates it) and a per-step reward of about
# 0.28, where the sampled-quadrotor evaluation pays about 1.29 per step and no such penalty: the two evaluations do not
# share their MDP constants, so nothing says they share the termination threshold or the initial distribution either.
REFERENCE_LOG_CRAZYFLIE = {"share_terminated": 0.05, "episode_length": 477.4, "episode_length_std": 98.4,
                           "share_terminated_last20": 0.0425, "episode_length_last20": 481.1,
                           "terminated_episode_length_implied": 48.0, "return_mean": 127.8, "return_std": 69.4}


def _nominal_crazyflie(O, weights, n=16384, seed=3, **over):
    cfg = O.default_config()
    cfg.domain_randomization = 0
    for k, v in over.items():
        setattr(cfg, k, v)
    P = O.sample_initial_parameters(cfg, seed, 0, 0, n)
    st = O.Stats(n)
    S = O.sample_initial_state(cfg, seed, st.episode, 0, P)
    H = np.zeros((n, 16), np.float32)
    O.rollout(cfg, weights, seed, 0, 0, P, S, H, 500, 0, st, O.max_threads())
    assert (st.fin_counts == 1).all()
    term = st.fin_terminated.astype(bool)
    L = st.fin_lengths.astype(np.float64)
    return term.mean(), L.mean(), (L[term].mean() if term.any() else float("nan")), L.std()


def test_nominal_crazyflie_statistics_of_this_specification(oracle, weights):
    """What THIS specification gives for the nominal Crazyflie ('stated mismatch'): about one
    episode in a hundred terminates - the log says one in twenty - while the terminated episodes end after ~52 steps,
    as the log implies (48).  The failures are the right kind (hard initial conditions lost within half a second);
    there are five times too few of them.  Pinned here so that a change of the specification shows."""
    share, length, len_term, _ = _nominal_crazyflie(oracle, weights)
    assert 0.005 < share < 0.016, share
    assert 493.0 < length < 498.0, length
    assert abs(len_term - REFERENCE_LOG_CRAZYFLIE["terminated_episode_length_implied"]) < 12.0, len_term


@pytest.mark.xfail(strict=True, reason="stated mismatch (DESIGN.md section 2): the log's nominal-Crazyflie evaluation "
                                       "terminates 5 % of its episodes, this specification 1 %; strict, so that a "
                                       "specification change that closes the gap is noticed and documented")
def test_nominal_crazyflie_statistics_against_the_reference_training_log(oracle, weights):
    share, length, _, _ = _nominal_crazyflie(oracle, weights)
    assert abs(share - REFERENCE_LOG_CRAZYFLIE["share_terminated"]) < 0.012
    assert abs(length - REFERENCE_LOG_CRAZYFLIE["episode_length"]) < 5.0


@pytest.mark.parametrize("candidate", [dict(init_max_angle=1.9), dict(termination_position=0.75), dict(disturbance_force_std=0.19)])
def test_single_constant_candidates_that_would_close_the_crazyflie_gap(oracle, weights, candidate):
    """Each of three different single-constant changes reproduces the log's nominal-Crazyflie share and length (an
    initial tilt of up to 109 degrees instead of 90 even its length spread, 98.4 against the log's 98.4): two logged
    numbers cannot choose between them, and each of them moves the sampled-quadrotor statistic the default threshold
    was fitted to away from the log (0.042 -> 0.07 ... 0.09, measured) unless the domain-randomisation ranges - this
    repository's own - move too.  None is adopted; the degeneracy is what is recorded."""
    share, length, _, _ = _nominal_crazyflie(oracle, weights, **candidate)
    assert abs(share - REFERENCE_LOG_CRAZYFLIE["share_terminated"]) < 0.012, (candidate, share)
    assert abs(length - REFERENCE_LOG_CRAZYFLIE["episode_length"]) < 5.0, (candidate, length)


def test_action_history_raw_or_clipped_hardly_moves_the_statistics(oracle, weights):
    """rq_env_config.action_history_raw (which action ActionHistory(1) keeps is not in the reference tree): the
    closed-loop statistics do not depend on it within their sampling error - the shipped policy saturates its
    commands on a few transient steps only.  Measured (16 384 envs x 2 seeds): nominal 0.0093 -> 0.0091 terminated,
    sampled quadrotors 0.0398 -> 0.0378."""
    a = _nominal_crazyflie(oracle, weights, n=8192)
    b = _nominal_crazyflie(oracle, weights, n=8192, action_history_raw=1)
    assert abs(a[0] - b[0]) < 0.004 and abs(a[1] - b[1]) < 2.0, (a, b)
    # and the switch does what it says: the observation carries the unclipped command
    cfg = oracle.default_config()
    cfg.action_history_raw = 1
    P = oracle.sample_initial_parameters(cfg, 0, 0, 0, 4)
    S = oracle.sample_initial_state(cfg, 0, np.zeros(4, np.uint32), 0, P)
    act = np.tile(np.array([[2.5, -3.0, 0.25, 1.0]], np.float32), (4, 1))
    S1, _, _ = oracle.step(cfg, P, S, act)
    assert np.array_equal(oracle.observe(cfg, 0, 0, 0, P, S1)[:, 18:22], act)
    cfg.action_history_raw = 0
    S0, _, _ = oracle.step(cfg, P, S, act)
    assert np.array_equal(oracle.observe(cfg, 0, 0, 0, P, S0)[:, 18:22], np.clip(act, -1, 1))
    assert np.array_equal(S0[:, :17], S1[:, :17])          # the dynamics see the clipped command either way
