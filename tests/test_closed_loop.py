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
# (tags crazyflie/*).  Round 4: tests/golden/reference_log.json (make_reference_log.py) holds all eleven tags of the log.
# One epoch of this tag is an evaluation of 100 episodes (every logged share is a multiple of 0.01), so the LAST record -
# 0.05 / 477.4, what rounds 2 and 3 compared with - carries a sampling error of 0.02; the pool of the last 100 epochs
# (10 000 episodes of nearly the final policy) says 0.034 +- 0.002 terminated, 485.0 steps, and hence terminated episodes
# that end after (485.0 - 0.966 * 500) / 0.034 = 57.1 +- 0.7 steps.  crazyflie/return/* are not comparable: that
# environment pays 0.29 per step and carries a termination penalty of 100 ... 200 (first epoch: return -102 +- 5.8 at length
# 29.7 +- 8.3 with every episode terminated; regression of return on length and share over the last 500 epochs:
# 0.289 per step, -200 per unit share), where the sampled-quadrotor evaluation pays 1.30 per step: the two evaluations do
# not share their reward constants, so nothing says they share the termination thresholds or the initial distribution.
import json as _json
import os as _os

with open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "reference_log.json")) as _fh:
    REFERENCE_LOG_FULL = _json.load(_fh)
_CF_POOL = REFERENCE_LOG_FULL["pooled"]["crazyflie"]["last_100"]
REFERENCE_LOG_CRAZYFLIE = {"share_terminated": _CF_POOL["share_terminated"], "episode_length": _CF_POOL["episode_length"],
                           "terminated_episode_length_implied": _CF_POOL["terminated_episode_length"],
                           "share_terminated_last_record": REFERENCE_LOG_FULL["tags"]["crazyflie/share_terminated"]["last"],
                           "episode_length_last_record": REFERENCE_LOG_FULL["tags"]["crazyflie/episode_length/mean"]["last"]}


def test_reference_log_fixture_holds_every_tag_of_the_training_log():
    """The fixture against what DESIGN.md and the tests quote from it (and against itself)."""
    log = REFERENCE_LOG_FULL
    assert len(log["tags"]) == 11 and log["tags"]["loss"]["n"] == 146103
    assert log["episodes_per_evaluation"] == {"crazyflie": 100, "evaluation": 10000}
    assert log["actor_meta"]["environment"]["observation"] == \
        "Position.OrientationRotationMatrix.LinearVelocity.AngularVelocityDelayed(0).ActionHistory(1)"
    t = log["tags"]
    assert t["crazyflie/share_terminated"]["last"] == pytest.approx(0.05) and t["crazyflie/episode_length/mean"]["last"] == pytest.approx(477.44)
    assert t["crazyflie/return/mean"]["first"] == pytest.approx(-102.0, abs=0.01) and t["crazyflie/episode_length/mean"]["first"] == pytest.approx(29.72)
    assert t["evaluation/share_terminated"]["last"] == pytest.approx(REFERENCE_LOG["share_terminated"], abs=5e-4)
    assert t["evaluation/episode_length/mean"]["last"] == pytest.approx(REFERENCE_LOG["episode_length"], abs=0.05)
    for pre in ("crazyflie", "evaluation"):      # series and summaries agree; every share is a multiple of 1 / episodes
        v = np.array(log["series"][pre + "/share_terminated"]["value"])
        assert v[-1] == pytest.approx(t[pre + "/share_terminated"]["last"]) and len(v) == t[pre + "/share_terminated"]["n"]
        k = v * log["episodes_per_evaluation"][pre]
        assert np.abs(k - np.round(k)).max() < 0.02          # float32 shares
    assert 56.0 < _CF_POOL["terminated_episode_length"] < 58.5 and 0.030 < _CF_POOL["share_terminated"] < 0.038
    assert log["return_regression_last_500"]["crazyflie"]["reward_per_step"] == pytest.approx(0.289, abs=0.005)
    assert log["return_regression_last_500"]["evaluation"]["reward_per_step"] == pytest.approx(1.299, abs=0.005)


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
    episode in a hundred terminates - the log's last 100 epochs say 3.4 in a hundred - while the terminated episodes end
    after ~52 steps, as the log implies (57).  The failures are the right kind (hard initial conditions lost within half
    a second); there are three times too few of them.  Pinned here so that a change of the specification shows."""
    share, length, len_term, _ = _nominal_crazyflie(oracle, weights)
    assert 0.005 < share < 0.016, share
    assert 493.0 < length < 498.0, length
    assert abs(len_term - REFERENCE_LOG_CRAZYFLIE["terminated_episode_length_implied"]) < 9.0, len_term


def test_the_reference_log_is_not_reproduced_and_the_record_says_so():
    """Round 6 closed the topic: tools/policy_competence.py mapped where the shipped policy is competent over
    scale, thrust-to-weight, motor time constants, torque constant and inertia, read candidate training ranges off that boundary by
    a rule fixed before the comparison, and ran the log's two evaluations with them and NO fitted constant
    (profiles/r06_policy_competence.json): neither share terminated nor time to failure of `evaluation/*` or `crazyflie/*` falls
    out.  The strict xfail rounds 3 - 5 kept on the crazyflie share (log 0.034, this specification 0.010) is retired with it: the
    mismatch is a stated property of this specification, pinned by test_nominal_crazyflie_statistics_of_this_specification above,
    and only the l2f sources - absent from the reference tree - can remove it."""
    with open(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "profiles", "r06_policy_competence.json")) as fh:
        rec = _json.load(fh)
    assert rec["verdict"].startswith("NOT REPRODUCED")
    p = rec["predictions"]["no_fitted_constant_3m"]
    for key in ("evaluation_competence_ranges", "evaluation_specification_ranges", "crazyflie_nominal"):
        assert not (p[key]["fits_log"]["share"] and p[key]["fits_log"]["time_to_failure"]), key
    assert rec["log"]["crazyflie"]["share_terminated"] == pytest.approx(REFERENCE_LOG_CRAZYFLIE["share_terminated"])
    # the rule was applied as written: every range is a run of competent cells around the centre
    for ax, r in rec["competence_ranges"].items():
        one = rec["one_at_a_time"][ax]
        inside = [c for v, c in zip(one["values"], one["competent"]) if r["range"][0] <= v <= r["range"][1]]
        assert inside and all(inside), ax
        assert r["range"][0] <= rec["centre"][ax] <= r["range"][1]


def test_policy_competence_boundary_of_record(oracle, weights):
    """Three cells of profiles/r06_policy_competence.json re-measured (same builder, same rule, fewer envs): the shipped policy
    hovers the centre cell to 1.5 cm, loses vehicles below a thrust-to-weight of ~1.5 and cannot hold position beyond a torque
    constant of ~0.03 s - where this specification's randomisation ranges end."""
    import sys
    sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools"))
    import policy_competence as pc
    cfg = oracle.default_config()
    cfg.termination_position = pc.LOST_M
    rows = {}
    for name, change in (("centre", {}), ("low_thrust", dict(t2w=1.3396)), ("high_kq", dict(kq=0.0528))):
        cell = dict(pc.CENTRE)
        cell.update(change)
        rows[name] = pc.run(oracle, weights, cfg, pc.params_cell(1024, **cell), 11)
    assert pc.competent(rows["centre"], 0.0) and rows["centre"]["steady_state_p_median"] < 0.02
    assert not pc.competent(rows["low_thrust"], 0.0) and 0.2 < rows["low_thrust"]["share_terminated"] < 0.45
    assert not pc.competent(rows["high_kq"], 0.0) and rows["high_kq"]["share_terminated"] > 0.7
    # the cell builder against the specification's own sampler: the nominal Crazyflie differs from the centre cell only in its
    # thrust constant and rotor speed range (same thrust-to-weight to 0.2 %)
    cfg.domain_randomization = 0
    P0 = oracle.sample_initial_parameters(cfg, 0, 0, 0, 1)[0]
    Pc = pc.params_cell(1, **pc.CENTRE)[0]
    assert np.allclose(Pc[:16], P0[:16], rtol=1e-6) and np.allclose(Pc[19:22], [0.006, 0.15, 0.15], rtol=1e-6)
    t2w = lambda P: 4 * P[18] * P[23] ** 2 / (P[0] * 9.81)
    assert abs(t2w(Pc) - 2.25) < 1e-3 and abs(t2w(P0) - 2.25) < 0.01


# tolerance of the two crazyflie/* comparisons below: three standard errors of the log's pool plus this sample's own
_TOL_SHARE = 3 * _CF_POOL["share_terminated_se"] + 0.004
_TOL_AFTER = 3 * _CF_POOL["terminated_episode_length_se"] + 3.5


@pytest.mark.parametrize("candidate,share_fits,after_fits", [
    (dict(init_max_angle=1.83), True, True),              # more failures of the same kind: survives
    (dict(disturbance_force_std=0.16), True, True),       # likewise
    (dict(termination_position=0.8), True, False),        # failures that end sooner: the share fits, the time to failure does not
    (dict(init_max_position=0.8), False, False),
    (dict(termination_angular_velocity=10.0), False, False),
])
def test_time_to_failure_discriminates_the_candidates_for_the_crazyflie_gap(oracle, weights, candidate, share_fits, after_fits):
    """Round 4: the log's pooled crazyflie/* tags give a THIRD number - terminated episodes end after 57.1 +- 0.7 steps -
    and it separates the single-constant candidates of round 3 (tools/env_constraint_study.py,
    profiles/r04_env_constraints.json): a larger initial tilt (about 105 degrees) and a per-episode force disturbance (about
    0.16 m g) reproduce share AND time to failure; a tighter position threshold, a wider initial position or an
    angular-velocity threshold end their failures after 36 - 39 (13) steps and are rejected.  Neither survivor is adopted:
    each moves the sampled-quadrotor statistic (evaluation/*: 0.0425) to 0.066 under this repository's own randomisation
    ranges, which the tree does not state either - the specification stays under-determined by the tree (DESIGN.md
    section 2) and what is pinned here is which constants the log can and cannot tell apart."""
    share, length, len_term, _ = _nominal_crazyflie(oracle, weights, **candidate)
    assert (abs(share - REFERENCE_LOG_CRAZYFLIE["share_terminated"]) <= _TOL_SHARE) == share_fits, (candidate, share)
    assert (abs(len_term - REFERENCE_LOG_CRAZYFLIE["terminated_episode_length_implied"]) <= _TOL_AFTER) == after_fits, (candidate, len_term)


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
