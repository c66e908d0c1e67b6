"""parameter / state sampling, observe, step (reward, termination masks, counters) against the oracle -
bit-exact - plus closed forms of the physics on the GPU kernels and the closed-loop statistics on the GPU.

Parity of the HIP path (through the C ABI of libraptor_quad.so) against the oracle.  Bars (DESIGN.md "Parity"):
  * integer / index / mask work, parameter sampling, observe (no noise) and env transitions for identical inputs: BIT-EXACT;
  * anything behind a transcendental (actor gates, sin/cos of the initial attitude, Box-Muller noise): float32 tolerance stated per test;
  * the actor additionally against the reference's own known-answer vectors (< 1e-5).
(Round 6 split tests/test_gpu_parity.py - 2 987 lines, one module - by SURVEY.md section 8 row group, so that a red run names its row.)
"""
import os

import numpy as np
import pytest

from gpu_common import INIT_TOL, NOISE_TOL, World      # noqa: F401

pytestmark = pytest.mark.gpu

# ------------------------------------------------------------------------------ sampling ---
@pytest.mark.parametrize("dr", [0, 1])
def test_sample_initial_parameters_bit_exact(device, oracle, dr):
    w = World(device, oracle, 1000, seed=3, offset=12345, domain_randomization=dr)
    assert np.array_equal(w.params.numpy(), w.P)


def test_sample_initial_state(device, oracle):
    w = World(device, oracle, 1000, seed=4, disturbance_force_std=0.05, disturbance_torque_std=0.01)
    S = w.state.numpy()
    assert S.shape == (1000, 27)
    # positions, velocities, rotor speeds, action history: no transcendental -> bit-exact
    for sl in (slice(0, 3), slice(7, 13), slice(13, 21)):
        assert np.array_equal(S[:, sl], w.S[:, sl])
    assert np.abs(S[:, 3:7] - w.S[:, 3:7]).max() < INIT_TOL
    scale = np.abs(w.S[:, 21:27]).max(axis=0) + 1e-30
    assert (np.abs(S[:, 21:27] - w.S[:, 21:27]) / scale).max() < 1e-4     # Box-Muller
    # second call = next episode, again in agreement
    w.vector.sample_initial_state(device, w.env, w.params, w.state, w.rng)
    S2 = oracle.sample_initial_state(w.cfg, 4, w.st.episode, 0, w.P)
    assert np.array_equal(w.state.numpy()[:, 0:3], S2[:, 0:3]) and not np.array_equal(S2[:, 0:3], S[:, 0:3])


# ------------------------------------------------------------------------------ observe ----
def test_observe_bit_exact(w1k):
    w = w1k
    w.sync_oracle_to_gpu_state()
    obs = np.zeros((w.n, 26), np.float32)
    w.vector.observe(w.device, w.env, w.params, w.state, obs, w.rng)
    assert np.array_equal(obs, w.O.observe(w.cfg, w.seed, 0, w.offset, w.P, w.S))
    # device-resident variant holds the same values
    w.vector.observe(w.device, w.env, w.params, w.state, None, w.rng)
    assert np.array_equal(w.env.observation(), obs)


def test_observe_with_noise(device, oracle):
    w = World(device, oracle, 1000, seed=9, noise_position=0.1, noise_orientation=0.02,
              noise_linear_velocity=0.3, noise_angular_velocity=0.4)
    w.sync_oracle_to_gpu_state()
    for epoch in range(3):
        obs = np.zeros((w.n, 26), np.float32)
        w.vector.observe(device, w.env, w.params, w.state, obs, w.rng)
        ref = oracle.observe(w.cfg, 9, epoch, 0, w.P, w.S)
        std = np.array([0.1] * 3 + [0.02] * 9 + [0.3] * 3 + [0.4] * 3 + [1] * 8, np.float32)
        assert (np.abs(obs - ref) / std).max() < NOISE_TOL * 10
        assert np.array_equal(obs[:, 18:], ref[:, 18:])
    assert w.rng.epoch == 3


# ------------------------------------------------------------------------------ step -------
def test_step_bit_exact_with_fed_actions(device, oracle):
    """README loop with identical actions on both sides: 100 transitions, every bit equal."""
    w = World(device, oracle, 1000, seed=1, disturbance_force_std=0.05, disturbance_torque_std=0.01)
    w.sync_oracle_to_gpu_state()
    rng = np.random.default_rng(0)
    for t in range(100):
        act = rng.uniform(-1.5, 1.5, (w.n, 4)).astype(np.float32)
        dts = w.vector.step(device, w.env, w.params, w.state, act, w.next_state, w.rng)
        w.state.assign(w.next_state)
        ns, r, term = oracle.step(w.cfg, w.P, w.S, act)
        oracle.stats_update(w.cfg, r, term, w.st)
        w.S = ns
        if t % 25 == 0 or t == 99:
            assert np.array_equal(w.state.numpy(), ns), t
            assert np.array_equal(w.env.rewards(), r) and np.array_equal(w.env.terminated(), term)
            assert np.array_equal(w.env.returns(), w.st.returns)
            assert np.array_equal(w.env.episode_steps(), w.st.steps)
    assert len(dts) == w.n and dts[-1] == pytest.approx(0.01)
    assert np.array_equal(w.env.finished_counts(), w.st.fin_counts)


def test_step_termination_masks_and_nan(device, oracle):
    w = World(device, oracle, 64, domain_randomization=0, init_guidance=1.0)
    S = w.state.numpy()
    S[1, 0], S[1, 7] = 2.999, 5.0
    S[2, 10] = np.nan
    S[3, 9] = 2000.0
    w.state.set(S)
    act = np.tile(w.P[:, 25:26], (1, 4)).astype(np.float32)
    act[5] = np.nan
    w.vector.step(device, w.env, w.params, w.state, act, w.next_state, w.rng)
    ns, r, term = oracle.step(w.cfg, w.P, S, act)
    assert term[:6].tolist() == [0, 1, 1, 1, 0, 0]
    assert np.array_equal(w.env.terminated(), term)
    assert np.array_equal(w.env.rewards(), r, equal_nan=True)
    assert np.array_equal(w.next_state.numpy(), ns, equal_nan=True)
    assert np.array_equal(w.env.finished_terminated(), term.astype(np.uint32))


def test_step_termination_flag_every_component_nonfinite_or_over_threshold(device, oracle):
    """The HIP step derives `terminated` from NaN-propagating group maxima (v_maximum3_f32); the oracle
    tests every component on its own.  One env per (component, poison) pair: NaN, +inf, -inf, a huge finite
    value and values just under / over each threshold, alone and next to a NaN neighbour."""
    poisons = [np.nan, np.inf, -np.inf, 3.0e38, -3.0e38]
    cases = [(f, v) for f in range(17) for v in poisons]
    thr = {0: 3.0, 1: 3.0, 2: 3.0, 7: 1000.0, 8: 1000.0, 9: 1000.0, 10: 1000.0, 11: 1000.0, 12: 1000.0}
    n = 64 * ((len(cases) + 4 * len(thr) + 63) // 64)
    w = World(device, oracle, n, domain_randomization=0, init_guidance=1.0)
    cfg = w.cfg
    thr = {f: (cfg.termination_position if f < 3 else cfg.termination_linear_velocity if f < 10
               else cfg.termination_angular_velocity) for f in thr}
    S = w.state.numpy()
    S[:, 0:3] = 0.0; S[:, 7:13] = 0.0          # hover at the origin: one step moves nothing past a threshold
    e = 0
    for f, v in cases:
        S[e, f] = v; e += 1
    for f, t in thr.items():                   # threshold edges: x(t+dt) = x(t) + O(dt) for these fields
        for scale, nan_neighbour in ((0.9, False), (1.1, False), (1.1, True), (-1.1, False)):
            S[e, f] = t * scale
            if nan_neighbour:
                S[e, f + 1 if f % 3 != 2 and f != 12 else f - 1] = np.nan
            e += 1
    w.state.set(S)
    act = np.tile(w.P[:, 25:26], (1, 4)).astype(np.float32)
    w.vector.step(device, w.env, w.params, w.state, act, w.next_state, w.rng)
    ns, r, term = oracle.step(w.cfg, w.P, S, act)
    assert 0 < term[:e].sum() and term[e:].sum() == 0
    assert np.array_equal(w.env.terminated(), term)
    assert np.array_equal(w.env.rewards(), r, equal_nan=True)
    assert np.array_equal(w.next_state.numpy(), ns, equal_nan=True)


def test_step_in_place_equals_out_of_place(device, oracle):
    w = World(device, oracle, 300, seed=2)
    act = np.random.default_rng(1).uniform(-1, 1, (300, 4)).astype(np.float32)
    w.vector.step(device, w.env, w.params, w.state, act, w.next_state, w.rng)
    out = w.next_state.numpy()
    w.vector.step(device, w.env, w.params, w.state, act, w.state, w.rng)
    assert np.array_equal(w.state.numpy(), out)


@pytest.mark.parametrize("case", range(12))
def test_randomised_configs_step_observe_bit_exact(device, oracle, case):
    """Fuzz over the MDP configuration (dt, gravity, limits, reward weights, thresholds, disturbances,
    batch size incl. ragged tails): params / observe / 30 chained transitions stay bit-identical."""
    r = np.random.default_rng(1000 + case)
    n = int(r.choice([1, 7, 64, 65, 129, 640, 1000, 4097]))
    over = dict(dt=float(r.choice([0.002, 0.005, 0.01, 0.02])), gravity=float(r.uniform(1.0, 12.0)),
                episode_step_limit=int(r.integers(3, 40)), domain_randomization=int(r.integers(0, 2)),
                dr_scale_min=float(r.uniform(0.4, 1.0)), dr_scale_max=float(r.uniform(1.5, 9.0)),
                init_guidance=float(r.uniform(0, 1)), init_max_position=float(r.uniform(0.1, 2.0)),
                disturbance_force_std=float(r.choice([0.0, 0.05])), disturbance_torque_std=float(r.choice([0.0, 0.02])),
                reward_scale=float(r.uniform(0.1, 2)), reward_constant=float(r.uniform(0, 2)),
                reward_termination_penalty=float(r.uniform(-5, 0)), reward_action=float(r.uniform(0, 1)),
                termination_enabled=int(r.integers(0, 2)), termination_position=float(r.uniform(0.3, 3.0)),
                termination_linear_velocity=float(r.uniform(1.0, 100.0)),
                termination_angular_velocity=float(r.uniform(5.0, 100.0)))
    w = World(device, oracle, n, seed=int(r.integers(0, 2 ** 40)), offset=int(r.integers(0, 2 ** 34)), **over)
    assert np.array_equal(w.params.numpy(), w.P)
    w.sync_oracle_to_gpu_state()
    obs = np.zeros((n, 26), np.float32)
    for t in range(30):
        w.vector.observe(device, w.env, w.params, w.state, obs, w.rng)
        assert np.array_equal(obs, oracle.observe(w.cfg, w.seed, t, w.offset, w.P, w.S))
        act = r.uniform(-1.3, 1.3, (n, 4)).astype(np.float32)
        w.vector.step(device, w.env, w.params, w.state, act, w.state, w.rng)
        w.S, rew, term = oracle.step(w.cfg, w.P, w.S, act)
        oracle.stats_update(w.cfg, rew, term, w.st)
        assert np.array_equal(w.state.numpy(), w.S, equal_nan=True)
        assert np.array_equal(w.env.rewards(), rew, equal_nan=True) and np.array_equal(w.env.terminated(), term)
    assert np.array_equal(w.env.finished_counts(), w.st.fin_counts)
    assert np.array_equal(w.env.finished_lengths(), w.st.fin_lengths)
    assert np.array_equal(w.env.finished_returns(), w.st.fin_returns, equal_nan=True)


def test_policy_stabilises_gpu_simulation(device, oracle):
    """The functional pin of the conventions, on the HIP path itself."""
    w = World(device, oracle, 4096, seed=5, termination_enabled=1)
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 500, "fused", False)
    term = w.env.finished_terminated()
    assert (w.env.finished_counts() == 1).all()
    assert term.mean() < 0.07
    S = w.state.numpy()
    assert np.median(np.linalg.norm(S[term == 0, :3], axis=1)) < 0.1


@pytest.mark.parametrize("mode", ["fused", "chained"])
def test_done_codes_frozen_and_episode_index_getters(device, oracle, mode):
    """Freeze mode: after an episode of 20 steps every env is frozen with done code 2 (step limit) or 1
    (terminated), further steps leave them at 4; sample_initial_state unfreezes and advances the episode index.
    Auto-reset: nothing freezes, the episode index counts the resets."""
    n = 300
    w = World(device, oracle, n, seed=51, episode_step_limit=20)
    assert not w.env.frozen().any() and (w.env.episode_index() == 1).all()
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 20, mode, False)
    codes = w.env.done_codes()
    assert w.env.frozen().all() and set(np.unique(codes)) <= {1, 2} and (codes == 2).sum() > n // 2
    assert np.array_equal(codes == 1, w.env.terminated() == 1)
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 3, mode, False)
    assert (w.env.done_codes() == 4).all()
    w.vector.sample_initial_state(device, w.env, w.params, w.state, w.rng)
    assert not w.env.frozen().any() and (w.env.episode_index() == 2).all()
    w.policy.reset()
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 45, mode, True)
    assert not w.env.frozen().any()
    assert np.array_equal(w.env.episode_index(), 2 + w.env.finished_counts() - 1)


def test_closed_loop_statistics_against_the_reference_training_log_on_the_gpu(device, oracle):
    """The HIP path's own closed-loop statistics against numbers the reference produced (its training log, see
    tests/test_closed_loop.py::REFERENCE_LOG): 65 536 randomised quadrotors, shipped policy, position termination
    threshold 1 m -> share of terminated episodes 0.042 +- 0.008 and mean episode length 482.8 +- 4."""
    from test_closed_loop import REFERENCE_LOG
    w = World(device, oracle, 65536, seed=7, termination_position=1.0)
    w.policy.reset()
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 500, "fused", autoreset=False)
    assert (w.env.finished_counts() == 1).all()
    share, length = w.env.finished_terminated().mean(), w.env.finished_lengths().mean()
    assert abs(share - REFERENCE_LOG["share_terminated"]) < 0.008, share
    assert abs(length - REFERENCE_LOG["episode_length"]) < 4.0, length


def test_nominal_crazyflie_statistics_on_the_gpu(device, oracle):
    """The second record of the reference's log (tests/test_closed_loop.py::REFERENCE_LOG_CRAZYFLIE: the pool of the last
    100 epochs of the crazyflie/* tags, tests/golden/reference_log.json) on the HIP path: 65 536 nominal Crazyflies, shipped
    policy.  The specification's own figures (about 1 % terminated after ~52 steps; the log: 3.4 % after 57) - a stated
    mismatch, DESIGN.md section 2 - and the two single-constant candidates that survive the log's time-to-failure (round 4:
    initial tilt up to 1.83 rad, force disturbance 0.16 m g) as the HIP kernels compute them; a tighter position threshold
    reproduces the share and fails the time to failure here as in the oracle."""
    from test_closed_loop import REFERENCE_LOG_CRAZYFLIE as LOG

    def stats(**over):
        w = World(device, oracle, 65536, seed=3, domain_randomization=0, **over)
        w.policy.reset()
        w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 500, "fused", autoreset=False)
        assert (w.env.finished_counts() == 1).all()
        term = w.env.finished_terminated().astype(bool)
        L = w.env.finished_lengths().astype(np.float64)
        return term.mean(), L.mean(), L[term].mean()

    share, length, len_term = stats()
    print(f"[nominal Crazyflie, HIP path] share terminated {share:.4f} (log {LOG['share_terminated']}), length {length:.1f} "
          f"(log {LOG['episode_length']}), terminated after {len_term:.1f} steps (log implies {LOG['terminated_episode_length_implied']})")
    assert 0.006 < share < 0.014 and 494.0 < length < 497.5, (share, length)
    assert abs(len_term - LOG["terminated_episode_length_implied"]) < 9.0, len_term
    assert abs(share - LOG["share_terminated"]) > 0.015          # the stated mismatch, on this path as well
    for over in (dict(init_max_angle=1.83), dict(disturbance_force_std=0.16)):
        share, length, len_term = stats(**over)
        print(f"[nominal Crazyflie, HIP path, {over}] {share:.4f} / {length:.1f} / terminated after {len_term:.1f}")
        assert abs(share - LOG["share_terminated"]) < 0.009 and abs(length - LOG["episode_length"]) < 4.5, (over, share, length)
        assert abs(len_term - LOG["terminated_episode_length_implied"]) < 6.0, (over, len_term)
    share, length, len_term = stats(termination_position=0.8)
    assert abs(share - LOG["share_terminated"]) < 0.009 and abs(len_term - LOG["terminated_episode_length_implied"]) > 12.0, (share, len_term)


def test_action_history_raw_on_the_gpu(device, oracle):
    """rq_env_config.action_history_raw: k_step and the fused kernel keep the policy's raw output as ActionHistory(1),
    bit for bit what the oracle keeps; the dynamics still see the clipped command."""
    w = World(device, oracle, 777, seed=5, action_history_raw=1)
    rng = np.random.default_rng(0)
    act = (rng.standard_normal((w.n, 4)) * 2.0).astype(np.float32)
    w.sync_oracle_to_gpu_state()
    w.vector.step(device, w.env, w.params, w.state, act, w.next_state, w.rng)
    ns, r, t = oracle.step(w.cfg, w.P, w.S, act)
    got = w.next_state.numpy()
    assert np.array_equal(got, ns) and np.array_equal(got[:, 17:21], act)
    assert np.array_equal(w.env.rewards(), r)
    # fused == chained with the switch on, over saturating steps
    a = World(device, oracle, 3000, seed=6, action_history_raw=1, init_max_angle=3.0)
    b = World(device, oracle, 3000, seed=6, action_history_raw=1, init_max_angle=3.0)
    a.vector.rollout(device, a.env, a.params, a.state, a.policy, a.rng, 40, "fused", True)
    b.vector.rollout(device, b.env, b.params, b.state, b.policy, b.rng, 40, "chained", True)
    sa, sb = a.state.numpy(), b.state.numpy()
    assert np.array_equal(sa, sb)
    assert np.abs(sa[:, 17:21]).max() > 1.0          # the history really holds unclipped commands


def test_closed_form_physics_at_full_size(device):
    """Size-independent properties of the env step that no restatement is needed for, on 262 144 domain-randomised
    envs on the GPU (BASELINE config 3's batch): with the rotors' thrust switched off the body is in free fall and
    torque-free, so after K steps
      * v = v0 - g K dt z, p = p0 + v0 K dt - g (K dt)^2 / 2 z      (RK4 is exact for a quadratic),
      * the world-frame angular momentum R(q) J w and the rotational energy w.Jw/2 are conserved (J is not
        isotropic: the body precesses, only a correct quaternion / Euler integration keeps both),
      * |q| = 1,
      * each rotor speed follows the first-order lag towards its set-point with RK4's own amplification factor
        rho(x) = 1 - x + x^2/2 - x^3/6 + x^4/24, x = dt / T, per step.
    These are checks of k_step against closed forms, not against oracle/."""
    import torch
    import raptor_amd.l2f as l2f
    n, K = 262144, 200
    v = l2f.VectorModule(n, 0)
    rng, env, params, state = v.VectorRng(), v.VectorEnvironment(), v.VectorParameters(), v.VectorState()
    v.initialize_rng(device, rng, 123)
    v.initialize_environment(device, env)
    cfg = env.config
    cfg.termination_enabled = 0
    cfg.disturbance_force_std = 0.0
    cfg.disturbance_torque_std = 0.0
    cfg.init_max_angular_velocity = 6.0
    cfg.episode_step_limit = 10 * K
    env.config = cfg
    v.sample_initial_parameters(device, env, params, rng)
    v.sample_initial_state(device, env, params, state, rng)
    device.synchronize()
    P, S = params.tensor(), state.tensor()
    P[16:19] = 0.0                                            # T = c0 + c1 r + c2 r^2 = 0: no force, no torque
    g_ = torch.Generator(device="cuda").manual_seed(7)
    act = torch.rand(4, P.shape[1], device="cuda", generator=g_) * 2.4 - 1.2       # some outside [-1, 1]: clipped
    env.action_tensor().copy_(act)
    torch.cuda.synchronize()
    s0 = S[:, :n].double().clone()
    p64 = P[:, :n].double()
    for _ in range(K):
        v.step(device, env, params, state, None, state, rng)
    device.synchronize()
    s1 = S[:, :n].double()
    dt, g = float(cfg.dt), float(cfg.gravity)
    T = K * dt

    def rot(q):                                               # body -> world, q = (w, x, y, z)
        w, x, y, z = q
        return torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)]),
                            torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)]),
                            torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)])])

    # free fall
    v_expect = s0[7:10].clone(); v_expect[2] -= g * T
    p_expect = s0[0:3] + s0[7:10] * T; p_expect[2] -= 0.5 * g * T * T
    dv = (s1[7:10] - v_expect).abs().max().item()
    dp = (s1[0:3] - p_expect).abs().max().item()
    # torque-free rotation
    J = p64[1:4]
    L0 = torch.einsum("ijn,jn->in", rot(s0[3:7]), J * s0[10:13])
    L1 = torch.einsum("ijn,jn->in", rot(s1[3:7]), J * s1[10:13])
    dL = ((L1 - L0).norm(dim=0) / L0.norm(dim=0).clamp_min(1e-12)).max().item()
    E0, E1 = (J * s0[10:13] ** 2).sum(0), (J * s1[10:13] ** 2).sum(0)
    dE = ((E1 - E0).abs() / E0.clamp_min(1e-30)).max().item()
    dq = (s1[3:7].norm(dim=0) - 1).abs().max().item()
    moved = (s1[10:13] - s0[10:13]).abs().max().item()        # the precession is real: w itself changes
    # rotors
    sp = p64[22] + (act[:, :n].double().clamp(-1, 1) + 1) * 0.5 * (p64[23] - p64[22])
    tau = torch.where(sp > s0[13:17], p64[20].expand(4, n), p64[21].expand(4, n))
    x = dt / tau
    rho = 1 - x + x ** 2 / 2 - x ** 3 / 6 + x ** 4 / 24
    r_expect = sp + (s0[13:17] - sp) * rho ** K
    dr = ((s1[13:17] - r_expect).abs() / p64[23]).max().item()
    print(f"\n[closed forms, {n} envs x {K} steps] max |dv| {dv:.2e} m/s, |dp| {dp:.2e} m, angular momentum {dL:.2e} rel, "
          f"rotational energy {dE:.2e} rel, | |q| - 1 | {dq:.2e}, rotor speed {dr:.2e} of rpm_max; w moved by {moved:.2f} rad/s")
    assert dv < 5e-4 and dp < 2e-3, (dv, dp)
    assert dL < 1e-4 and dE < 1e-4, (dL, dE)
    assert dq < 5e-6, dq
    assert moved > 0.5
    assert dr < 2e-5, dr
    assert torch.equal(S[17:21, :n], act[:, :n].clamp(-1, 1))           # ActionHistory(1) = the clipped action
    # the same body under the FUSED rollout kernel (hand-packed env step, policy in the loop: with the thrust off its
    # actions only move the rotors): same start, same closed forms
    from raptor_amd.foundation_policy import Raptor
    S[:, :n] = s0.float()
    torch.cuda.synchronize()
    policy = Raptor(device)
    policy.reset()
    v.rollout(device, env, params, state, policy, rng, K, "fused", autoreset=False)
    device.synchronize()
    s2 = S[:, :n].double()
    dv2 = (s2[7:10] - v_expect).abs().max().item()
    dp2 = (s2[0:3] - p_expect).abs().max().item()
    L2 = torch.einsum("ijn,jn->in", rot(s2[3:7]), J * s2[10:13])
    dL2 = ((L2 - L0).norm(dim=0) / L0.norm(dim=0).clamp_min(1e-12)).max().item()
    dE2 = (((J * s2[10:13] ** 2).sum(0) - E0).abs() / E0.clamp_min(1e-30)).max().item()
    dq2 = (s2[3:7].norm(dim=0) - 1).abs().max().item()
    print(f"[closed forms, fused rollout] max |dv| {dv2:.2e} m/s, |dp| {dp2:.2e} m, angular momentum {dL2:.2e} rel, "
          f"rotational energy {dE2:.2e} rel, | |q| - 1 | {dq2:.2e}")
    assert dv2 < 5e-4 and dp2 < 2e-3 and dL2 < 1e-4 and dE2 < 1e-4 and dq2 < 5e-6, (dv2, dp2, dL2, dE2, dq2)
    # k_step and the fused kernel run the same arithmetic: position, attitude, velocities agree bit for bit
    assert torch.equal(S[0:13, :n].double(), s1[0:13])


def test_hover_equilibrium_and_torque_sign_conventions(device):
    """The conventions /root/reference/README.md:23-27 states (FLU body frame, motor order front-right, back-right,
    back-left, front-left, actions in [-1, 1]) checked on the GPU env step without any restatement: a level body at
    its hover rotor speed with the hover action stays put; more thrust on the right pair (y < 0) lifts the right side
    (rotation about -x), on the back pair pitches the nose down (+y), on the back-right / front-left pair (the +z
    reaction torques) yaws left (+z)."""
    import torch
    import raptor_amd.l2f as l2f
    n = 4096
    v = l2f.VectorModule(n, 0)
    rng, env, params, state = v.VectorRng(), v.VectorEnvironment(), v.VectorParameters(), v.VectorState()
    v.initialize_rng(device, rng, 5)
    v.initialize_environment(device, env)
    cfg = env.config
    cfg.termination_enabled = 0
    cfg.disturbance_force_std = 0.0
    cfg.disturbance_torque_std = 0.0
    env.config = cfg
    v.sample_initial_parameters(device, env, params, rng)          # domain-randomised: every env its own body
    v.sample_initial_state(device, env, params, state, rng)
    device.synchronize()
    P, S, A = params.tensor(), state.tensor(), env.action_tensor()
    level = torch.zeros(27, n, device="cuda")
    level[3] = 1.0                                                   # q = identity
    level[13:17] = P[24, :n]                                         # hover rotor speed
    hover = P[25, :n]

    def run(action, steps):
        S[:, :n] = level
        A[:, :n] = action
        torch.cuda.synchronize()
        for _ in range(steps):
            v.step(device, env, params, state, None, state, rng)
        device.synchronize()
        return S[:, :n].double()

    s = run(hover.expand(4, n), 100)                                 # 1 s of hover
    drift, speed, spin = s[0:3].abs().max().item(), s[7:10].abs().max().item(), s[10:13].abs().max().item()
    print(f"\n[hover, {n} randomised bodies, 100 steps] |p| {drift:.2e} m, |v| {speed:.2e} m/s, |w| {spin:.2e} rad/s")
    assert drift < 2e-4 and speed < 5e-4 and spin < 1e-4
    up = 0.2
    for name, rotors, axis, sign in (("roll", (0, 1), 10, -1.0), ("pitch", (1, 2), 11, 1.0), ("yaw", (1, 3), 12, 1.0)):
        a = hover.expand(4, n).clone()
        for r in range(4):
            a[r] += up if r in rotors else -up
        s = run(a, 5)
        turn = sign * s[axis]
        others = [k for k in (10, 11, 12) if k != axis]
        assert (turn > 0).all(), name                               # the named axis turns the stated way
        assert s[others[0]].abs().max() < 1e-3 * turn.min() and s[others[1]].abs().max() < 1e-3 * turn.min(), name


def test_env_spec_fixture_on_the_gpu(device):
    """The committed spec-freeze fixture (tests/golden/env_spec.npz: parameters, states, actions of a recorded
    closed loop; DESIGN.md section 2 says what it is and is not) through the HIP kernels: every recorded transition
    (state_k, action_k) -> (state_k+1, reward, terminated) and every observation, bit for bit; the actor's actions
    within its tolerance.  No oracle code runs in this test."""
    import raptor_amd.l2f as l2f
    from raptor_amd.foundation_policy import Raptor
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "env_spec.npz"))
    n, steps, seed, offset = (int(x) for x in g["meta"])
    v = l2f.VectorModule(n, offset)
    rng, env, params, state, nxt = v.VectorRng(), v.VectorEnvironment(), v.VectorParameters(), v.VectorState(), v.VectorState()
    v.initialize_rng(device, rng, seed)
    v.initialize_environment(device, env)
    assert bytes(env.config) == g["config_bytes"].tobytes()            # the default MDP is part of the specification
    v.sample_initial_parameters(device, env, params, rng)
    assert np.array_equal(params.numpy(), g["params"])
    v.sample_initial_state(device, env, params, state, rng)
    assert np.allclose(state.numpy(), g["state0"], rtol=0, atol=2e-6)  # sinf/cosf of the initial attitude
    policy = Raptor(device)
    policy.reset()
    prev = g["state0"]
    worst = 0.0
    for k in range(steps):
        state.set(prev)
        obs = np.zeros((n, 26), np.float32)
        v.observe(device, env, params, state, obs, rng)
        assert np.array_equal(obs, g["obs"][k]), k
        act = policy.evaluate_step(obs[:, :22])
        worst = max(worst, float(np.abs(act - g["act"][k]).max()))
        v.step(device, env, params, state, g["act"][k], nxt, rng)
        assert np.array_equal(nxt.numpy(), g["state"][k]), k
        assert np.array_equal(env.rewards(), g["reward"][k]) and np.array_equal(env.terminated(), g["terminated"][k]), k
        prev = g["state"][k]
    assert worst < 1e-5, worst
