"""The loop body x K in one launch (north_star, SURVEY.md section 2.2 k_rollout_fused): fused == chained bit for bit, determinism, the long
closed loop against the oracle, BASELINE configs 3 and 4 (sharding invariance).

Parity of the HIP path (through the C ABI of libraptor_quad.so) against the oracle.  Bars (DESIGN.md "Parity"):
  * integer / index / mask work, parameter sampling, observe (no noise) and env transitions for identical inputs: BIT-EXACT;
  * anything behind a transcendental (actor gates, sin/cos of the initial attitude, Box-Muller noise): float32 tolerance stated per test;
  * the actor additionally against the reference's own known-answer vectors (< 1e-5).
(Round 6 split tests/test_gpu_parity.py - 2 987 lines, one module - by SURVEY.md section 8 row group, so that a red run names its row.)
"""
import os

import numpy as np
import pytest

from gpu_common import ACTOR_TOL, INIT_TOL, World, _closed_loop_agreement, _lib_set_epoch      # noqa: F401

pytestmark = pytest.mark.gpu

def test_bf16_fused_equals_chained(device, oracle):
    a = World(device, oracle, 300, seed=8, episode_step_limit=40)
    b = World(device, oracle, 300, seed=8, episode_step_limit=40)
    a.policy.set_precision("bf16"); b.policy.set_precision("bf16")
    a.vector.rollout(device, a.env, a.params, a.state, a.policy, a.rng, 100, "fused", True)
    b.vector.rollout(device, b.env, b.params, b.state, b.policy, b.rng, 100, "chained", True)
    assert np.array_equal(a.state.numpy(), b.state.numpy())
    assert np.array_equal(a.policy.hidden_state(300), b.policy.hidden_state(300))


def test_device_resident_chain_equals_host_chain(device, oracle):
    """observe(None) -> evaluate_step_device -> step(None) == the NumPy-passing loop, bit for bit."""
    a = World(device, oracle, 500, seed=6)
    b = World(device, oracle, 500, seed=6)
    obs = np.zeros((500, 26), np.float32)
    a.policy.reset(); b.policy.reset()
    for _ in range(20):
        a.vector.observe(device, a.env, a.params, a.state, obs, a.rng)
        act = a.policy.evaluate_step(obs[:, :22])
        a.vector.step(device, a.env, a.params, a.state, act, a.next_state, a.rng)
        a.state.assign(a.next_state)
        b.vector.observe(device, b.env, b.params, b.state, None, b.rng)
        b.policy.evaluate_step_device(b.env)
        b.vector.step_device(device, b.env, b.params, b.state, b.state, b.rng)
    assert np.array_equal(a.state.numpy(), b.state.numpy())
    assert np.array_equal(a.env.returns(), b.env.returns())


@pytest.mark.parametrize("precision", ["fp32", "bf16", "f16x2"])
@pytest.mark.parametrize("n", [1, 65, 4097, 70001])
def test_rollout_fused_equals_chained_ragged_sizes_all_precisions(device, oracle, n, precision):
    """One env, one lane past a wave, one past a 4 096-env block, and a batch past 65 536 (where the fused kernel
    switches to its two-waves-per-SIMD build): the fused kernel and the chain of API-granular kernels share one
    actor step function and one env step function per precision, so they agree bit for bit - tail lanes, the LDS
    tile of a partly filled wave and the mailbox path of the small batches included."""
    kw = dict(seed=5, episode_step_limit=9)
    a, b = World(device, oracle, n, **kw), World(device, oracle, n, **kw)
    a.policy.set_precision(precision); b.policy.set_precision(precision)
    for chunk in (7, 12):
        a.vector.rollout(device, a.env, a.params, a.state, a.policy, a.rng, chunk, "fused", True)
        b.vector.rollout(device, b.env, b.params, b.state, b.policy, b.rng, chunk, "chained", True)
    assert np.array_equal(a.state.numpy(), b.state.numpy())
    assert np.array_equal(a.policy.hidden_state(n), b.policy.hidden_state(n))
    assert np.array_equal(a.env.returns(), b.env.returns()) and np.array_equal(a.env.finished_counts(), b.env.finished_counts())
    assert a.env.finished_counts().min() >= 1                      # episodes ended and restarted on the way


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_streaming_actor_step_past_the_batch_where_waves_take_several_groups(device, oracle, precision):
    """Round 4: from 262 144 envs on k_actor_step is another instantiation - a wave streams through groups / 1 024 groups
    of 64 envs with the next group's inputs in flight.  262 144 + 129 envs (4 groups per wave, the last wave's groups
    partly and wholly past the batch): the chain built on it equals the fused kernel bit for bit, and the actions of a
    sample of envs equal the oracle's actor on the same observations."""
    n = 262144 + 129
    kw = dict(seed=9, episode_step_limit=4)
    a, b = World(device, oracle, n, **kw), World(device, oracle, n, **kw)
    a.policy.set_precision(precision); b.policy.set_precision(precision)
    for chunk in (3, 2):
        a.vector.rollout(device, a.env, a.params, a.state, a.policy, a.rng, chunk, "fused", True)
        b.vector.rollout(device, b.env, b.params, b.state, b.policy, b.rng, chunk, "chained", True)
    assert np.array_equal(a.state.numpy(), b.state.numpy())
    assert np.array_equal(a.policy.hidden_state(n), b.policy.hidden_state(n))
    assert np.array_equal(a.env.returns(), b.env.returns())
    # without auto-reset: envs freeze on the way (their stores are the ones the streaming kernel sends out of range)
    for chunk in (2, 4):
        a.vector.rollout(device, a.env, a.params, a.state, a.policy, a.rng, chunk, "fused", False)
        b.vector.rollout(device, b.env, b.params, b.state, b.policy, b.rng, chunk, "chained", False)
    assert a.env.frozen().all() and np.array_equal(a.env.frozen(), b.env.frozen())
    assert np.array_equal(a.state.numpy(), b.state.numpy())
    assert np.array_equal(a.policy.hidden_state(n), b.policy.hidden_state(n))
    assert np.array_equal(a.env.done_codes(), b.env.done_codes())
    for w in (a, b):                                     # thawed again by the next auto-reset launch
        w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 1, "fused" if w is a else "chained", True)
    assert np.array_equal(a.state.numpy(), b.state.numpy())
    if precision == "fp32":
        b.vector.observe(device, b.env, b.params, b.state, None, b.rng)
        obs = b.env.observation()
        H = b.policy.hidden_state(n)
        b.policy.evaluate_step_device(b.env)
        act = b.env.action()
        pick = np.r_[0:70, 131000:131100, n - 200:n]
        Hs = np.ascontiguousarray(H[pick])
        ref = oracle.actor_batch_step(b.policy.weights, np.ascontiguousarray(obs[pick, :22]), Hs)
        assert np.abs(act[pick] - ref).max() < 10 * ACTOR_TOL
        assert np.abs(b.policy.hidden_state(n)[pick] - Hs).max() < 10 * ACTOR_TOL


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_short_launches_follow_every_change_between_them(device, oracle, precision):
    """Many short fused launches with episode ends everywhere (the ahead-of-time sampled next-episode values are parked, used
    and refilled across them), and between the launches everything those values are a function of is changed in turn -
    another seed, another initial-state range, a disturbance switched on, parameters re-sampled and set from the host, the
    episode counters moved by sample_initial_state and by a chained rollout, a frozen batch thawed, another parameter
    object: after every change the fused path equals the chain of API-granular kernels bit for bit (state, policy state,
    every statistic).  (Written for an experiment that kept the parked values from launch to launch - profiles/
    r04_ab_not_kept.txt; what it pins holds for any such cache.)"""
    n = 777
    kw = dict(seed=21, episode_step_limit=5, termination_position=0.6)
    a, b = World(device, oracle, n, **kw), World(device, oracle, n, **kw)
    a.policy.set_precision(precision); b.policy.set_precision(precision)

    def both(f):
        f(a); f(b)

    def run(chunks, autoreset=True, modes=("fused", "chained")):
        for c in chunks:
            a.vector.rollout(device, a.env, a.params, a.state, a.policy, a.rng, c, modes[0], autoreset)
            b.vector.rollout(device, b.env, b.params, b.state, b.policy, b.rng, c, modes[1], autoreset)
        assert np.array_equal(a.state.numpy(), b.state.numpy())
        assert np.array_equal(a.policy.hidden_state(n), b.policy.hidden_state(n))
        for name in ("returns", "episode_steps", "finished_returns", "finished_lengths", "finished_counts",
                     "finished_terminated", "rewards", "terminated", "done_codes", "frozen", "episode_index"):
            assert np.array_equal(getattr(a.env, name)(), getattr(b.env, name)()), name

    def set_cfg(w, **over):
        cfg = w.env.config
        for k, v in over.items():
            setattr(cfg, k, v)
        w.env.config = cfg

    run([1, 1, 2, 1, 3, 1, 1, 7, 1, 2])                                   # values parked by one launch, used by the next
    both(lambda w: w.vector.initialize_rng(device, w.rng, 99))             # another seed
    run([1, 2, 1, 1, 4])
    both(lambda w: set_cfg(w, init_max_position=0.2, init_max_angle=0.3))  # another initial-state distribution
    run([1, 1, 3, 1])
    both(lambda w: set_cfg(w, disturbance_force_std=0.1, disturbance_torque_std=0.05))     # values 13..18 come alive
    run([2, 1, 1, 5])
    both(lambda w: w.vector.sample_initial_parameters(device, w.env, w.params, w.rng))     # mass / arm scale the disturbance
    run([1, 1, 2, 1])
    P = a.params.numpy().copy()
    P[:, 0] *= np.float32(1.25)                                                               # heavier: set from the host
    both(lambda w: w.params.set(P))
    run([1, 3, 1, 1])
    both(lambda w: w.vector.sample_initial_state(device, w.env, w.params, w.state, w.rng))   # moves every episode counter
    run([1, 1, 2])
    run([3, 4], modes=("chained", "chained"))                                                # counters moved by the other path
    run([1, 1, 1, 6])
    run([9], autoreset=False)                                                                 # every env ends and freezes ...
    assert a.env.frozen().all()
    run([1, 2, 1])                                                                            # ... and is thawed by the next launch
    for w in (a, b):                                                                          # another parameter OBJECT on the same env
        w.params = w.vector.VectorParameters()
        w.vector.sample_initial_parameters(device, w.env, w.params, w.rng)
    run([1, 1, 2, 1])


@pytest.mark.parametrize("precision", ["fp32", "bf16", "f16x2"])
def test_fused_rollout_is_deterministic(device, oracle, precision):
    """The same rollout twice gives the same bits - every build of the fused kernel that ships: the fp32 build with one wave per
    SIMD (4 097, 65 536 envs) and with two (70 001, 131 072, 262 144 + 129), the bf16 and split-f16 builds (one wave per SIMD at
    every size since round 5), with and without auto-reset, short launches and a longer one - AND the SampleAndSquash
    instantiations of each (round 4's verdict: the builds behind that stage had no determinism test).  Round 4 found a
    two-waves-per-SIMD bf16 build differing FROM RUN TO RUN (lanes 48 .. 63 of ~1 % of the waves) when compiled with the max-ilp
    instruction scheduler; round 5 found the cause - a gfx950 fault of one packed-fp32 op_sel form beside another wave's 16-bit MFMA
    (profiles/r05_bf16_two_wave_hunt.md; raptor_amd/gfx950_errata.py rewrites the form away) - and took that build, the slower one
    anyway, out of the product.  Three repetitions per case; the chained path the same."""
    cases = [(4097, 3, True, "off"), (65536, 2, True, "off"), (70001, 3, True, "off"), (131072, 3, True, "off"), (131072, 3, False, "off"),
             (131072, 40, True, "off"), (262144 + 129, 2, True, "off"),
             (65536, 3, True, "mean"), (131072, 3, True, "mean"), (131072, 3, False, "sample"), (70001, 25, True, "sample")]
    for n, steps, autoreset, sas in cases:
        for rep in range(3):
            kw = dict(seed=40 + rep, episode_step_limit=4 if steps < 10 else 25)
            a, b = World(device, oracle, n, **kw), World(device, oracle, n, **kw)
            for w in (a, b):
                w.policy.set_precision(precision)
                if sas != "off":
                    w.policy.set_sample_and_squash(sas, log_std_bias=np.full(4, -1.0, np.float32), seed=5)
                w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, steps, "fused", autoreset)
            assert np.array_equal(a.state.numpy(), b.state.numpy()), (n, steps, autoreset, sas, rep)
            assert np.array_equal(a.policy.hidden_state(n), b.policy.hidden_state(n)), (n, steps, autoreset, sas, rep)
            assert np.array_equal(a.env.returns(), b.env.returns())
    a, b = World(device, oracle, 131072, seed=3, episode_step_limit=4), World(device, oracle, 131072, seed=3, episode_step_limit=4)
    a.policy.set_precision(precision); b.policy.set_precision(precision)
    for w in (a, b):
        w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 5, "chained", True)
    assert np.array_equal(a.state.numpy(), b.state.numpy()) and np.array_equal(a.policy.hidden_state(131072), b.policy.hidden_state(131072))


@pytest.mark.parametrize("case", range(int(os.environ.get("RQ_RANDOM_CASES", "32"))))
def test_fused_equals_chained_over_random_settings(device, oracle, case):
    """Random batch size, episode limit, thresholds, noise, disturbance, action history, actor precision, recording and chunking -
    every one with many episode ends per env (the fused kernel's ahead-of-time sampling, its episode-end records written
    from inside the loop and its rare-path addressing are what this is after): state, policy state and all episode
    statistics agree bit for bit with the chain of API-granular kernels."""
    r = np.random.default_rng(1000 + case)
    n = int(r.choice([1, 63, 64, 65, 777, 4097, 20000]))
    kw = dict(seed=int(r.integers(1, 1000)), episode_step_limit=int(r.integers(3, 60)),
              termination_position=float(r.choice([0.2, 0.5, 1.0])))
    if r.random() < 0.5:
        kw.update(noise_position=0.01, noise_angular_velocity=0.05)
    if r.random() < 0.5:
        kw.update(disturbance_force_std=0.0, disturbance_torque_std=0.0)
    if r.random() < 0.25:
        kw.update(action_history_raw=1)
    autoreset = bool(r.random() < 0.8)
    precision = str(r.choice(["fp32", "fp32", "bf16", "f16x2"]))
    record = bool(r.random() < 0.4)
    chunks = [int(r.choice([1, 2, 3, 7, 20, 61, 150])) for _ in range(int(r.integers(2, 6)))]
    a, b = World(device, oracle, n, **kw), World(device, oracle, n, **kw)
    a.policy.set_precision(precision); b.policy.set_precision(precision)
    ta = a.vector.Trajectory(a.env, sum(chunks)) if record else None
    tb = b.vector.Trajectory(b.env, sum(chunks)) if record else None
    total = 0
    for chunk in chunks:
        a.vector.rollout(device, a.env, a.params, a.state, a.policy, a.rng, chunk, "fused", autoreset, trajectory=ta)
        b.vector.rollout(device, b.env, b.params, b.state, b.policy, b.rng, chunk, "chained", autoreset, trajectory=tb)
        total += chunk
    kw = dict(kw, precision=precision, record=record, autoreset=autoreset, chunks=chunks)
    assert np.array_equal(a.state.numpy(), b.state.numpy()), (n, kw, total)
    if record:
        A, B = ta.numpy(), tb.numpy()
        assert np.array_equal(A["done"], B["done"]), (n, kw)
        live = A["done"] != 4
        for key in ("obs", "act", "rew"):
            assert np.array_equal(A[key][live], B[key][live]), (key, n, kw)
    assert np.array_equal(a.policy.hidden_state(n), b.policy.hidden_state(n))
    for name in ("returns", "episode_steps", "finished_returns", "finished_lengths", "finished_counts",
                 "finished_terminated", "rewards", "terminated", "done_codes", "frozen", "episode_index"):
        assert np.array_equal(getattr(a.env, name)(), getattr(b.env, name)()), (name, n, kw, total)


def test_kernel_level_timing_records_and_leaves_results_alone(device, oracle):
    """rq_device_set_rollout_timing / rq_device_last_rollout_ms / rq_device_last_rollout_waves / rq_device_last_rollout_clock:
    every wave of a timed fused rollout leaves four ticks in order (in <= first step <= last step done <= out) and the die it ran on; the duration is
    plausible; the rollout's results are those of an untimed one."""
    n = 70001                                        # the two-waves-per-SIMD build; 1 094 waves
    a, b = World(device, oracle, n, seed=3), World(device, oracle, n, seed=3)
    device.set_rollout_timing(True)
    try:
        a.vector.rollout(device, a.env, a.params, a.state, a.policy, a.rng, 20, "fused", True)
        ms = device.last_rollout_ms()
        t_in, t_out, xcd, t_first, t_last = device.last_rollout_waves()
        ghz = device.last_rollout_clock_ghz()
    finally:
        device.set_rollout_timing(False)
    assert 1.2 < ghz < 2.6, ghz                     # the core clock the waves' steps ran at (the peak assumes 2.4 GHz)
    b.vector.rollout(device, b.env, b.params, b.state, b.policy, b.rng, 20, "fused", True)
    assert np.array_equal(a.state.numpy(), b.state.numpy())
    assert len(t_in) == (n + 63) // 64
    assert np.all(t_in <= t_first) and np.all(t_first <= t_last) and np.all(t_last <= t_out)
    assert xcd.min() >= 0 and xcd.max() <= 7 and len(np.unique(xcd)) == 8
    assert 0.02 < ms < 2.0, ms
    per_wave_us = (t_out - t_in).astype(np.float64) / 100.0          # 100 MHz ticks
    assert per_wave_us.max() <= ms * 1e3 + 0.5 and per_wave_us.min() > 10.0


@pytest.mark.parametrize("autoreset", [False, True])
@pytest.mark.parametrize("noise", ["position", "orientation", "linear_velocity", "angular_velocity", "all"])
def test_first_fused_step_equals_chained_in_every_noise_build(device, oracle, noise, autoreset):
    """One step from a fresh state, fused against chained, in the kernel builds the other tests reach only after many
    steps.  The first step is the one that consumes what the fused kernel's prologue computes ahead (tile 0's recurrent
    accumulators, ActorF32T::prime): round 3 had a build - noise + auto-reset - in which a register move of those
    accumulators was scheduled behind the branch that follows the prologue, 4 wait states after the MFMA instead of 11,
    and the first step of the 16 envs of every wave's tile 0 was garbage while every later step was right."""
    groups = ["position", "orientation", "linear_velocity", "angular_velocity"] if noise == "all" else [noise]
    kw = {"noise_" + g: 0.01 for g in groups}
    for n in (64, 777):
        a = World(device, oracle, n, seed=8, episode_step_limit=40, **kw)
        b = World(device, oracle, n, seed=8, episode_step_limit=40, **kw)
        a.vector.rollout(device, a.env, a.params, a.state, a.policy, a.rng, 1, "fused", autoreset)
        b.vector.rollout(device, b.env, b.params, b.state, b.policy, b.rng, 1, "chained", autoreset)
        assert np.array_equal(a.state.numpy(), b.state.numpy()), n
        assert np.array_equal(a.policy.hidden_state(n), b.policy.hidden_state(n)), n


@pytest.mark.parametrize("autoreset", [False, True])
def test_rollout_fused_equals_chained_bit_exact(device, oracle, autoreset):
    kw = dict(seed=8, episode_step_limit=40, noise_position=0.01, noise_angular_velocity=0.05)
    a = World(device, oracle, 777, **kw)
    b = World(device, oracle, 777, **kw)
    for chunk in (30, 50, 45):
        a.vector.rollout(device, a.env, a.params, a.state, a.policy, a.rng, chunk, "fused", autoreset)
        b.vector.rollout(device, b.env, b.params, b.state, b.policy, b.rng, chunk, "chained", autoreset)
    assert np.array_equal(a.state.numpy(), b.state.numpy())
    assert np.array_equal(a.policy.hidden_state(777), b.policy.hidden_state(777))
    for name in ("returns", "episode_steps", "finished_returns", "finished_lengths", "finished_counts",
                 "finished_terminated", "rewards", "terminated", "done_codes", "frozen", "episode_index"):
        assert np.array_equal(getattr(a.env, name)(), getattr(b.env, name)()), name
    assert a.rng.epoch == b.rng.epoch == 125
    if autoreset:
        assert (a.env.finished_counts() >= 3).all()
    else:
        assert (a.env.finished_counts() == 1).all()


@pytest.mark.parametrize("mode", ["fused", "chained"])
@pytest.mark.parametrize("dr", [0, 1])
def test_rollout_vs_oracle_closed_loop(device, oracle, weights, mode, dr):
    """500 closed-loop steps, policy in the loop."""
    w = World(device, oracle, 512, seed=11, domain_randomization=dr)
    w.sync_oracle_to_gpu_state()
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 500, mode, False)
    sel = _closed_loop_agreement(w, weights, 500, 0)
    assert (w.env.finished_counts() == 1).all()
    assert np.quantile(np.abs(w.policy.hidden_state(512)[sel] - w.H[sel]).max(axis=1), 0.99) < 1e-2


def test_rollout_autoreset_vs_oracle(device, oracle, weights):
    w = World(device, oracle, 256, seed=12, episode_step_limit=60)
    w.sync_oracle_to_gpu_state()
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 200, "fused", True)
    sel = _closed_loop_agreement(w, weights, 200, 1, min_same_history=0.99, min_insensitive=0.97)   # measured 1.0 / 1.0
    assert np.array_equal(w.env.episode_steps()[sel], w.st.steps[sel])


@pytest.mark.parametrize("mode", ["fused", "chained"])
def test_rollout_two_steps_ahead_everywhere(device, oracle, weights, mode):
    """Teacher-forced closed loop: along a 300-step oracle trajectory, every 20 steps load the
    oracle's (state, hidden) into the GPU and advance 2 steps with the policy in the loop.
    No horizon for chaos to act on, so EVERY env must agree: floats within 1e-4 abs / 1e-4 rel
    (actor transcendental ulps only), termination masks exactly."""
    w = World(device, oracle, 640, seed=13, episode_step_limit=10 ** 6, noise_position=0.01,
              noise_linear_velocity=0.02)
    w.sync_oracle_to_gpu_state()
    for t in range(0, 300, 20):
        w.state.set(w.S)
        w.policy.set_hidden_state(w.H)
        w.env.reset_statistics()
        assert w.rng.epoch == t
        w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 2, mode, False)
        S2, H2 = w.S.copy(), w.H.copy()
        st2 = oracle.Stats(w.n)
        oracle.rollout(w.cfg, weights, 13, t, 0, w.P, S2, H2, 2, 0, st2, 8)
        G = w.state.numpy()
        live = st2.frozen == 0
        assert np.array_equal(w.env.terminated(), st2.last_terminated)
        scale = np.maximum(np.abs(S2[live, :17]), 1.0)
        assert (np.abs(G[live, :17] - S2[live, :17]) / scale).max() < 1e-4, t
        assert np.abs(w.policy.hidden_state(w.n)[live] - H2[live]).max() < 1e-5
        assert np.abs(w.env.rewards()[live] - st2.last_reward[live]).max() < 1e-4
        # advance the oracle trajectory by 20 steps (frozen envs stay where they are)
        oracle.rollout(w.cfg, weights, 13, t, 0, w.P, w.S, w.H, 20, 0, w.st, 8)
        _lib_set_epoch(w, t + 20)


@pytest.mark.parametrize("case", range(6))
def test_resampled_states_follow_the_oracle_through_many_episodes(device, oracle, weights, case):
    """Episode limits of 1 .. 4 steps over a 12-step recorded fused rollout: every env starts 3 .. 12 episodes, and the
    state an episode starts from depends on (seed, episode counter, global env id) only - not on the actor - so the
    observation recorded right after every episode end must be the oracle's: position and velocities bit for bit, the
    rotation matrix to the sin / cos tolerance.  This is the ahead-of-time sampler's episode counter (refill, take, refill
    again inside one launch and across launches) against the reference restatement, not against the chained kernels."""
    r = np.random.default_rng(50 + case)
    n = int(r.choice([64, 200, 777]))
    limit = int(r.integers(1, 5))
    seed = int(r.integers(1, 500))
    w = World(device, oracle, n, seed=seed, episode_step_limit=limit, termination_enabled=0)
    w.sync_oracle_to_gpu_state()
    chunks = [int(c) for c in r.choice([1, 2, 3, 4, 6], size=4)]
    T = sum(chunks)
    tr = w.vector.Trajectory(w.env, T)
    for c in chunks:
        w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, c, "fused", True, trajectory=tr)
    ref = oracle.rollout_record(w.cfg, weights, seed, 0, 0, w.P, w.S, w.H, T, 1, w.st, 4)
    G = tr.numpy()
    assert np.array_equal(G["done"], ref["done"])
    starts = np.nonzero(ref["done"][:-1, 0] >= 1)[0] + 1            # steps whose observation is of a fresh state
    assert len(starts) >= T // limit - 1
    for t in starts:
        assert np.array_equal(G["obs"][t][:, 0:3], ref["obs"][t][:, 0:3]), t          # position
        assert np.array_equal(G["obs"][t][:, 12:18], ref["obs"][t][:, 12:18]), t      # linear, angular velocity
        assert np.abs(G["obs"][t][:, 3:12] - ref["obs"][t][:, 3:12]).max() < INIT_TOL * 4, t
        assert np.array_equal(G["obs"][t][:, 18:22], np.zeros((n, 4), np.float32)), t  # previous action of a new episode
    assert np.array_equal(w.env.episode_index(), w.st.episode)


# ------------------------------------------------------------------------------ scale ------
def test_sharding_invariance_and_determinism_at_full_size(device, oracle):
    """65 536 envs (BASELINE config 2): one batch == two half batches with global offsets,
    bit for bit (RNG keyed by global env id), and a repeated run reproduces itself."""
    n = 65536
    kw = dict(seed=21)
    full = World(device, oracle, n, **kw)
    lo = World(device, oracle, n // 2, offset=0, **kw)
    hi = World(device, oracle, n // 2, offset=n // 2, **kw)
    for w in (full, lo, hi):
        w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 100, "fused", True)
    Sf = full.state.numpy()
    assert np.array_equal(Sf, np.concatenate([lo.state.numpy(), hi.state.numpy()]))
    assert np.array_equal(full.env.returns(), np.concatenate([lo.env.returns(), hi.env.returns()]))
    again = World(device, oracle, n, **kw)
    again.vector.rollout(device, again.env, again.params, again.state, again.policy, again.rng, 100, "fused", True)
    assert np.array_equal(Sf, again.state.numpy())
    # size-independent properties: unit quaternions, rotor speeds inside their limits
    q = Sf[:, 3:7]
    assert np.abs(np.linalg.norm(q, axis=1) - 1).max() < 1e-5
    P = full.params.numpy()
    assert (Sf[:, 13:17] >= P[:, 22:23]).all() and (Sf[:, 13:17] <= P[:, 23:24]).all()


def test_config3_262144_envs_domain_randomised(device, oracle):
    """BASELINE config 3: 262 144 envs with per-env randomised mass / inertia / thrust parameters.
    Parameters bit-exact vs the oracle at full size; rollout checked through size-independent
    properties and a strided sample of envs two steps ahead of the oracle."""
    n = 262144
    w = World(device, oracle, n, seed=41)
    P = w.params.numpy()
    assert np.array_equal(P, w.P)
    assert len(np.unique(P[:, 0])) > 0.9 * n                      # every env its own quadrotor
    w.sync_oracle_to_gpu_state()
    idx = np.arange(0, n, 509)
    Ps, Ss, Hs = np.ascontiguousarray(w.P[idx]), np.ascontiguousarray(w.S[idx]), np.zeros((len(idx), 16), np.float32)
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 2, "fused", True)
    G = w.state.numpy()
    # the oracle on the sampled envs (RNG keyed per env: run them one by one with their global ids)
    st = oracle.Stats(1)
    worst = 0.0
    for k in range(0, len(idx), 8):
        i = int(idx[k])
        s1, h1 = Ss[k:k + 1].copy(), Hs[k:k + 1].copy()
        st = oracle.Stats(1); st.episode[:] = 1
        oracle.rollout(w.cfg, w.policy.weights, 41, 0, i, Ps[k:k + 1], s1, h1, 2, 1, st, 1)
        worst = max(worst, (np.abs(G[i, :17] - s1[0, :17]) / np.maximum(np.abs(s1[0, :17]), 1.0)).max())
    assert worst < 1e-4, worst
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 498, "fused", True)
    G = w.state.numpy()
    assert np.isfinite(G).all()
    assert np.abs(np.linalg.norm(G[:, 3:7], axis=1) - 1).max() < 1e-5
    assert (G[:, 13:17] >= P[:, 22:23]).all() and (G[:, 13:17] <= P[:, 23:24]).all()
    assert (w.env.finished_counts() >= 1).all() and w.env.finished_terminated().sum() / w.env.finished_counts().sum() < 0.07


def test_config4_shard_of_2097152_equals_slice_of_full_batch(device, oracle):
    """BASELINE config 4: 2 097 152 envs sharded 8 x 262 144.  On one GPU: the shard a rank would
    own (global ids 3*262144 ...) must equal the same slice of the unsharded 2 097 152-env batch,
    bit for bit, and the all-gather layout (contiguous by global id) is what raptor_amd.distributed
    assumes."""
    from raptor_amd.distributed import shard_range
    total, world, rank = 2097152, 8, 3
    start, count = shard_range(total, world, rank)
    assert (start, count) == (3 * 262144, 262144)
    full = World.__new__(World)
    import raptor_amd.l2f as l2f
    from raptor_amd.foundation_policy import Raptor

    def make(n, offset):
        v = l2f.VectorModule(n, offset)
        rng, env, params, state = v.VectorRng(), v.VectorEnvironment(), v.VectorParameters(), v.VectorState()
        v.initialize_rng(device, rng, 77); v.initialize_environment(device, env)
        v.sample_initial_parameters(device, env, params, rng); v.sample_initial_state(device, env, params, state, rng)
        pol = Raptor(device)
        v.rollout(device, env, params, state, pol, rng, 60, "fused", True)
        return state.numpy(), env.returns(), env.finished_counts()
    S_full, R_full, C_full = make(total, 0)
    S_sh, R_sh, C_sh = make(count, start)
    assert np.array_equal(S_full[start:start + count], S_sh)
    assert np.array_equal(R_full[start:start + count], R_sh) and np.array_equal(C_full[start:start + count], C_sh)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "f16x2"])
@pytest.mark.parametrize("mode", ["fused", "chained"])
def test_a_diverged_env_stays_alone_in_a_rollout(device, oracle, precision, mode):
    """The same property through the rollout kernels: with termination off, one env of a batch is handed an infinite (then
    a NaN) position; every other env's state, policy state and statistics after the rollout are bit for bit those of the
    batch without it - fused and chained, with and without auto-reset, in every precision, at a batch with a ragged tail
    and past the size where waves share a SIMD."""
    for n in (200, 70001):
        for autoreset in (True, False):
            for bad, victim in ((np.inf, 21), (np.nan, 21), (np.inf, n - 1)):      # n - 1: the env the tail lanes shadow
                out = []
                for poisoned in (False, True):
                    w = World(device, oracle, n, seed=77, termination_enabled=0)
                    w.policy.set_precision(precision)
                    w.policy.reset()
                    if poisoned:
                        S = w.state.numpy()
                        S[victim, 1] = bad
                        w.state.set(S)
                    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 7, mode, autoreset)
                    out.append((w.state.numpy(), w.policy.hidden_state(n), w.env.returns()))
                others = np.ones(n, bool); others[victim] = False
                for clean, dirty in zip(*out):
                    assert np.array_equal(clean[others], dirty[others]), (n, autoreset, bad, victim)
