"""the actor kernels (fp32 exact-MFMA, bf16, split-f16, evaluate_sequence, the optional
Standardize / SampleAndSquash stages) against the reference's known-answer vectors and the oracle.

Parity of the HIP path (through the C ABI of libraptor_quad.so) against the oracle.  Bars (DESIGN.md "Parity"):
  * integer / index / mask work, parameter sampling, observe (no noise) and env transitions for identical inputs: BIT-EXACT;
  * anything behind a transcendental (actor gates, sin/cos of the initial attitude, Box-Muller noise): float32 tolerance stated per test;
  * the actor additionally against the reference's own known-answer vectors (< 1e-5).
(Round 6 split tests/test_gpu_parity.py - 2 987 lines, one module - by SURVEY.md section 8 row group, so that a red run names its row.)
"""
import os

import numpy as np
import pytest

from gpu_common import ACTOR_TOL, World, BF16_KAT_TOL      # noqa: F401

pytestmark = pytest.mark.gpu

# ------------------------------------------------------------------------------ actor ------
def test_actor_selftest_against_reference_kats(device, kat):
    from raptor_amd.foundation_policy import Raptor
    x, y = kat
    err = Raptor(device).selftest(x, y, tolerance=ACTOR_TOL)
    assert err < ACTOR_TOL


def test_actor_boot_selftest_first_5_steps(device, kat):
    """The embedded backend's boot test: TEST_SEQUENCE_LENGTH_ACTUAL = 5, batch 2 (README.md:136-139)."""
    from raptor_amd.foundation_policy import Raptor
    x, y = kat
    assert Raptor(device).selftest(x[:5], y[:5], tolerance=ACTOR_TOL) < ACTOR_TOL


def test_actor_evaluate_step_loop_matches_kat_and_oracle(device, oracle, weights, kat):
    from raptor_amd.foundation_policy import Raptor
    x, y = kat
    pol = Raptor(device)
    pol.reset()
    h = np.zeros((2, 16), np.float32)
    worst_kat = worst_orc = 0.0
    for t in range(500):
        a = pol.evaluate_step(x[t])
        worst_kat = max(worst_kat, np.abs(a - y[t]).max())
        worst_orc = max(worst_orc, np.abs(a - oracle.actor_batch_step(weights, x[t], h)).max())
    assert worst_kat < ACTOR_TOL and worst_orc < ACTOR_TOL
    assert np.abs(pol.hidden_state(2) - h).max() < ACTOR_TOL
    # reset() restores the initial hidden state: the first step repeats
    pol.reset()
    assert np.abs(pol.evaluate_step(x[0]) - y[0]).max() < ACTOR_TOL


@pytest.mark.parametrize("batch", [1, 63, 64, 65, 1000])
def test_actor_ragged_batches_and_strided_input(device, oracle, weights, batch):
    from raptor_amd.foundation_policy import Raptor
    rng = np.random.default_rng(batch)
    wide = rng.standard_normal((batch, 26)).astype(np.float32)
    pol = Raptor(device)
    pol.reset()
    h = np.zeros((batch, 16), np.float32)
    for _ in range(3):
        a = pol.evaluate_step(wide[:, :22])            # non-contiguous view, as README.md:97
        ref = oracle.actor_batch_step(weights, wide, h)
        assert a.shape == (batch, 4) and np.abs(a - ref).max() < ACTOR_TOL


def test_actor_batch_change_requires_reset(device):
    import raptor_amd.l2f as l2f
    from raptor_amd.foundation_policy import Raptor
    pol = Raptor(device)
    pol.evaluate_step(np.zeros((4, 22), np.float32))
    with pytest.raises(l2f.RaptorQuadError) as e:
        pol.evaluate_step(np.zeros((5, 22), np.float32))
    assert e.value.status == -5
    pol.reset()
    assert pol.evaluate_step(np.zeros((5, 22), np.float32)).shape == (5, 4)


def test_optional_standardize_and_squash_stages(device, oracle, weights):
    """A6 / A7 of SURVEY.md section 8(a): identity by default, and when enabled equal to the oracle's actor
    fed standardised inputs / followed by tanh (their l2f / rl-tools parity is unpinned)."""
    from raptor_amd.foundation_policy import Raptor
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((300, 22)) * 3 + 1).astype(np.float32)
    mean = rng.standard_normal(22).astype(np.float32)
    std = rng.uniform(0.5, 2.0, 22).astype(np.float32)
    pol = Raptor(device)
    pol.set_standardize(mean, std)
    pol.set_squash(True)
    pol.reset()
    h = np.zeros((300, 16), np.float32)
    for _ in range(3):
        a = pol.evaluate_step(x)
        ref = np.tanh(oracle.actor_batch_step(weights, ((x - mean) / std).astype(np.float32), h))
        assert np.abs(a - ref).max() < 2e-5 and np.abs(a).max() <= 1.0
    pol.set_standardize(None, None)
    pol.set_squash(False)
    pol.reset()
    h = np.zeros((300, 16), np.float32)
    assert np.abs(pol.evaluate_step(x) - oracle.actor_batch_step(weights, x, h)).max() < 5e-5   # |x| up to ~10


def _bf16(x):
    """round-to-nearest-even fp32 -> bf16 -> fp32 (numpy)"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def _actor_bf16_model(w, x, h):
    """The bf16 kernel's arithmetic in numpy: operands rounded to bf16, fp32 accumulate, fp32 gates."""
    W0, b0 = w[0:352].reshape(16, 22), w[352:368]
    Wi, Wh = w[368:1136].reshape(48, 16), w[1136:1904].reshape(48, 16)
    bi, bh, W2, b2 = w[1904:1952], w[1952:2000], w[2016:2080].reshape(4, 16), w[2080:2084]
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    y0 = np.maximum(_bf16(x[:, :22]) @ _bf16(W0).T + _bf16(b0), 0).astype(np.float32)
    # the gate rows are pre-scaled (r, z by -log2 e, n by -2 log2 e) BEFORE they are rounded to bf16 (pack_policy_bf16)
    k = np.concatenate([np.full(32, -1.4426950408889634, np.float32), np.full(16, -2.8853900817779268, np.float32)])[:, None]
    gi, gh = (_bf16(y0) @ _bf16(k * Wi).T) / k.T, (_bf16(h) @ _bf16(k * Wh).T) / k.T
    r = sig(gi[:, :16] + gh[:, :16] + bi[:16] + bh[:16])
    z = sig(gi[:, 16:32] + gh[:, 16:32] + bi[16:32] + bh[16:32])
    n = np.tanh(gi[:, 32:] + bi[32:] + r * (gh[:, 32:] + bh[32:]))
    hn = ((1 - z) * n + z * h).astype(np.float32)
    return (_bf16(hn) @ _bf16(W2).T + b2).astype(np.float32), hn


def test_bf16_actor_against_kats_and_bf16_model(device, weights, kat):
    """BASELINE config 5: bf16 operands on the MFMA.  Against the reference KATs within the bf16
    tolerance, and within fp32 round-off of a numpy model of the same bf16-operand arithmetic."""
    from raptor_amd.foundation_policy import Raptor
    x, y = kat
    pol = Raptor(device, precision="bf16")
    err = pol.selftest(x, y, tolerance=BF16_KAT_TOL)
    assert 1e-4 < err < BF16_KAT_TOL       # really bf16 (not silently fp32), and within tolerance
    pol.reset()
    h = np.zeros((2, 16), np.float32)
    worst = 0.0
    for t in range(200):
        a = pol.evaluate_step(x[t])
        ref, h = _actor_bf16_model(weights, x[t], h)
        worst = max(worst, np.abs(a - ref).max())
        h = pol.hidden_state(2)             # teacher-force the model with the kernel's hidden state
    assert worst < 2e-3, worst              # rounding-boundary flips of individual bf16 operands only


def test_bf16_closed_loop_action_deviation(device, oracle):
    """Config 5 report: along an fp32 closed-loop trajectory of 4 096 domain-randomised quadrotors,
    the bf16 actor (own hidden state, same observations) deviates from the fp32 actor by a bounded
    amount, and flying the bf16 policy itself keeps the fleet as stable as the fp32 one."""
    from raptor_amd.foundation_policy import Raptor
    w = World(device, oracle, 4096, seed=31)
    p16 = Raptor(device, precision="bf16")
    p16.reset(); w.policy.reset()
    obs = np.zeros((4096, 26), np.float32)
    devs = []
    for t in range(500):
        w.vector.observe(device, w.env, w.params, w.state, obs, w.rng)
        a32 = w.policy.evaluate_step(obs[:, :22])
        a16 = p16.evaluate_step(obs[:, :22])
        devs.append(np.abs(a16 - a32).max(axis=1))
        w.vector.step(device, w.env, w.params, w.state, a32, w.state, w.rng)
    devs = np.array(devs)
    calm = np.abs(w.state.numpy()[:, :3]).max(axis=1) < 1.0
    print(f"bf16 vs fp32 action deviation over 500 closed-loop steps: max {devs[:, calm].max():.4f} "
          f"mean {devs[:, calm].mean():.5f}")
    assert devs[:, calm].mean() < 1e-2 and np.quantile(devs[:, calm], 0.999) < 0.1
    # and the bf16 policy in the loop (fused rollout)
    b = World(device, oracle, 4096, seed=31)
    b.policy.set_precision("bf16")
    b.vector.rollout(device, b.env, b.params, b.state, b.policy, b.rng, 500, "fused", False)
    f = World(device, oracle, 4096, seed=31)
    f.vector.rollout(device, f.env, f.params, f.state, f.policy, f.rng, 500, "fused", False)
    t16, t32 = b.env.finished_terminated().mean(), f.env.finished_terminated().mean()
    assert abs(t16 - t32) < 0.02 and t16 < 0.07
    r16, r32 = b.env.finished_returns().mean(), f.env.finished_returns().mean()
    assert abs(r16 - r32) / r32 < 0.02


def test_split_f16_actor_meets_the_fp32_bar(device, weights, kat, oracle):
    """RQ_POLICY_F16X2_MFMA: every operand as two f16 pieces on the f16 MFMA.  It has to pass what the fp32 build
    passes - both reference known-answer vectors to 1e-5 over 500 recurrent steps, random batches against the fp32
    oracle - through every kernel that carries an actor (step, sequence, fused, chained, relabel)."""
    from raptor_amd.foundation_policy import Raptor
    x, y = kat
    pol = Raptor(device, precision="f16x2")
    err = pol.selftest(x, y, tolerance=ACTOR_TOL)
    ref32 = Raptor(device).selftest(x, y, tolerance=ACTOR_TOL)
    print(f"\n[split-f16 actor] known-answer max abs error {err:.2e} (exact-fp32 MFMA build: {ref32:.2e})")
    assert err < ACTOR_TOL
    # one launch over the whole [500, 2, 22] tensor
    pol.reset()
    seq = pol.evaluate_sequence(x)
    assert np.abs(seq - y).max() < ACTOR_TOL
    # random batch, ragged size, against the oracle's fp32 actor over 30 recurrent steps
    rng = np.random.default_rng(3)
    n = 1000
    pol.reset()
    H = np.zeros((n, 16), np.float32)
    worst = 0.0
    for _ in range(30):
        obs = rng.normal(0, 1.5, (n, 22)).astype(np.float32)
        a = pol.evaluate_step(obs)
        worst = max(worst, float(np.abs(a - oracle.actor_batch_step(weights, obs, H)).max()))
    assert worst < ACTOR_TOL, worst
    # fused == chained bit for bit (one step function), with auto-reset and recording
    a_ = World(device, oracle, 300, seed=8, episode_step_limit=40)
    b_ = World(device, oracle, 300, seed=8, episode_step_limit=40)
    a_.policy.set_precision("f16x2"); b_.policy.set_precision("f16x2")
    ta, tb = a_.vector.Trajectory(a_.env, 100), b_.vector.Trajectory(b_.env, 100)
    a_.vector.rollout(device, a_.env, a_.params, a_.state, a_.policy, a_.rng, 100, "fused", True, trajectory=ta)
    b_.vector.rollout(device, b_.env, b_.params, b_.state, b_.policy, b_.rng, 100, "chained", True, trajectory=tb)
    assert np.array_equal(a_.state.numpy(), b_.state.numpy())
    assert np.array_equal(a_.policy.hidden_state(300), b_.policy.hidden_state(300))
    ha, hb = ta.numpy(), tb.numpy()
    assert all(np.array_equal(ha[k], hb[k]) for k in ("obs", "act", "rew", "done"))
    # relabelling the recording with a policy of the same weights and precision reproduces its actions
    teacher = Raptor(device, precision="f16x2")
    teacher.reset()
    assert np.array_equal(ta.relabel(teacher), ha["act"])
    # closed loop: flying the split-f16 policy is indistinguishable from flying the fp32 one at this horizon
    c = World(device, oracle, 4096, seed=31)
    d = World(device, oracle, 4096, seed=31)
    c.policy.set_precision("f16x2")
    c.vector.rollout(device, c.env, c.params, c.state, c.policy, c.rng, 60, "fused", False)
    d.vector.rollout(device, d.env, d.params, d.state, d.policy, d.rng, 60, "fused", False)
    dev_ = np.abs(c.state.numpy()[:, :13] - d.state.numpy()[:, :13]).max(axis=1)
    print(f"[split-f16 actor] 60-step closed loop vs fp32 actor: median |dstate| {np.median(dev_):.2e}, 99% {np.quantile(dev_, 0.99):.2e}")
    assert np.median(dev_) < 1e-4


@pytest.mark.parametrize("precision", ["fp32", "bf16", "f16x2"])
def test_sequence_and_relabel_are_deterministic_at_large_batches(device, oracle, precision):
    """Raptor.evaluate_sequence and Trajectory.relabel above 65 536 envs - where the fp32 actor switches to its two-waves-per-SIMD
    build and where the bf16 one used to (round 4 shipped ActorBF16Lean there without a determinism test; round 5 runs the
    one-wave bf16 build at every size): the same call twice, the same bits, three repetitions."""
    import torch
    from raptor_amd.foundation_policy import Raptor
    n, steps = 70001, 6
    x = torch.randn(steps, n, 22, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(7))
    outs = []
    for rep in range(3):
        pol = Raptor(device, precision=precision)
        pol.reset()
        outs.append((pol.evaluate_sequence(x).cpu().numpy(), pol.hidden_state(n)))
    for o, h in outs[1:]:
        assert np.array_equal(o, outs[0][0]) and np.array_equal(h, outs[0][1])
    w = World(device, oracle, n, seed=13, episode_step_limit=4)
    w.policy.set_precision(precision)
    traj = w.vector.Trajectory(w.env, steps)
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, steps, "fused", True, trajectory=traj)
    labels = []
    for rep in range(3):
        teacher = Raptor(device, precision=precision)
        teacher.reset()
        labels.append(traj.relabel(teacher))
    assert np.array_equal(labels[0], labels[1]) and np.array_equal(labels[0], labels[2])
    assert np.array_equal(labels[0], traj.numpy()["act"])            # the recording policy's own actions come back


def test_evaluate_sequence_against_reference_kats(device, kat, weights, oracle):
    """Raptor.evaluate_sequence on the known-answer tensors in their own layout [500, 2, 22] -> [500, 2, 4]:
    one kernel launch, < 1e-5 from the reference's outputs, and bit-identical to 500 evaluate_step calls."""
    from raptor_amd.foundation_policy import Raptor
    x, y = kat
    p = Raptor(device)
    p.reset()
    a = p.evaluate_sequence(x)
    assert a.shape == y.shape and np.max(np.abs(a - y)) < ACTOR_TOL
    q = Raptor(device)
    q.reset()
    steps = np.stack([q.evaluate_step(x[t]) for t in range(x.shape[0])])
    assert np.array_equal(a, steps)
    assert np.array_equal(p.hidden_state(2), q.hidden_state(2))


@pytest.mark.parametrize("batch,stride", [(1, 22), (65, 26), (1000, 23), (4096, 22), (70000, 22)])
def test_evaluate_sequence_ragged_strided_and_carried(device, oracle, weights, batch, stride):
    """Ragged batches, even/odd row strides (columns >= 22 never read), hidden state carried across calls:
    two half sequences equal the whole one bit for bit; against the oracle within ACTOR_TOL."""
    from raptor_amd.foundation_policy import Raptor
    T = 24
    rng = np.random.default_rng(batch)
    wide = np.full((T, batch, stride), np.nan, np.float32)
    wide[:, :, :22] = rng.standard_normal((T, batch, 22)).astype(np.float32)
    p = Raptor(device)
    p.reset()
    whole = p.evaluate_sequence(wide)
    q = Raptor(device)
    q.reset()
    halves = np.concatenate([q.evaluate_sequence(wide[:10]), q.evaluate_sequence(wide[10:])])
    assert np.array_equal(whole, halves)
    ref = oracle.actor_sequence(weights, np.ascontiguousarray(wide[:, :, :22]))
    assert np.max(np.abs(whole - ref)) < ACTOR_TOL
    assert np.array_equal(p.hidden_state(batch), q.hidden_state(batch))


def test_evaluate_sequence_device_tensors_and_bf16(device, kat):
    import torch
    from raptor_amd.foundation_policy import Raptor
    x, y = kat
    p = Raptor(device)
    p.reset()
    xt = torch.from_numpy(x).to("cuda:0")
    at = p.evaluate_sequence(xt)
    assert at.is_cuda and np.max(np.abs(at.cpu().numpy() - y)) < ACTOR_TOL
    b = Raptor(device, precision="bf16")
    b.reset()
    assert np.max(np.abs(b.evaluate_sequence(x) - y)) < 5e-2


@pytest.mark.parametrize("precision", ["fp32", "bf16", "f16x2"])
def test_rows_of_a_batch_do_not_see_each_others_infinities(device, precision):
    """"Works on batches by default" (README.md:24) means row by row.  Rounds 2-4 ran the 16-bit actors' output layer as one
    MFMA per 16-env tile and ACCUMULATED the four tiles of a wave into one result - tile t's weight rows are zero outside
    rows 4t .. 4t+3, which is exact for finite operands and 0 x inf = NaN otherwise: an infinite observation in row 5 of a
    bf16 batch turned the actions of rows 21, 37 and 53 into NaN (round 4).  Every tile keeps its own accumulator now.
    Checked in every precision: a non-finite observation, a huge one, and a non-finite policy state in one row leave every
    other row's action and policy state bit for bit what they are without it (also with the SampleAndSquash stage on,
    whose log-std head is laid out the same way)."""
    from raptor_amd.foundation_policy import Raptor
    rng = np.random.default_rng(5)
    for n in (64, 1100):
        x = rng.standard_normal((n, 22)).astype(np.float32)
        h = (0.3 * rng.standard_normal((n, 16))).astype(np.float32)
        others = np.ones(n, bool); others[5] = False
        for sas in (False, True):
            pol = Raptor(device, precision=precision)
            w_ls = (0.1 * rng.standard_normal((4, 16))).astype(np.float32)

            def run(obs, hid):
                if sas:                                # (again every time: the same sampling steps in every run)
                    pol.set_sample_and_squash("sample", w_ls, np.full(4, -1.0, np.float32), seed=3)
                pol.reset()
                pol.evaluate_step(x)                   # sizes the policy
                pol.set_hidden_state(hid)
                a = pol.evaluate_step(obs)
                return a, pol.hidden_state(n)
            a0, h0 = run(x, h)
            for bad in (np.inf, -np.inf, np.nan, 1e30):
                y = x.copy(); y[5, 3] = bad
                a1, h1 = run(y, h)
                assert np.array_equal(a1[others], a0[others]) and np.array_equal(h1[others], h0[others]), (n, sas, bad, "observation")
                g = h.copy(); g[5, 7] = bad
                a2, h2 = run(x, g)
                assert np.array_equal(a2[others], a0[others]) and np.array_equal(h2[others], h0[others]), (n, sas, bad, "policy state")


def test_split_f16_actor_saturates_out_of_range_inputs(device, oracle, weights):
    """RQ_POLICY_F16X2_MFMA beyond the f16 range (|x| >= 65 520 converts to infinity, and infinity minus infinity in the
    residual would be NaN in the GRU state for good): observations and layer_0's output are saturated at +-65 504 before
    the split, so evaluate_step, evaluate_sequence and a fused rollout with termination switched off from a caller-set
    far-away state all stay finite - as the fp32 build does - and inputs just inside the range are still fp32-grade."""
    import torch
    from raptor_amd.foundation_policy import Raptor
    rng = np.random.default_rng(11)
    B = 512
    pol, ref = Raptor(device, precision="f16x2"), Raptor(device)
    obs = rng.standard_normal((B, 22)).astype(np.float32)
    obs[:, 0] = 1.0e6                      # a position a diverging env reaches with termination off
    obs[1::2, 13] = -3.0e9
    obs[::7, 5] = np.float32(65520.0)      # exactly where the f16 conversion turns infinite
    obs[::11, 17] = np.nan
    pol.reset()
    for _ in range(3):
        a = pol.evaluate_step(obs)
        assert np.isfinite(a).all()
    h = pol.hidden_state(B)
    assert np.isfinite(h).all() and np.abs(h).max() <= 1.0 + 1e-6
    # evaluate_sequence on a tensor with the same rows
    x = torch.from_numpy(np.nan_to_num(np.stack([obs] * 4), nan=7.0e4)).to(f"cuda:{torch.cuda.current_device()}")
    pol.reset()
    y = pol.evaluate_sequence(x)
    assert torch.isfinite(y).all()
    # just inside the range the split is exact to 2^-22 relative: fp32-grade against the fp32 build (gates saturated or not)
    near = rng.standard_normal((B, 22)).astype(np.float32)
    near[:, 3] = 6.0e4
    pol.reset(); ref.reset()
    d = np.abs(pol.evaluate_step(near) - ref.evaluate_step(near)).max()
    assert d < 5e-3, d                     # operands of 6e4 carry 6e4 x 2^-22 = 0.014 absolute into the pre-activations
    # a fused rollout, termination off, from a state set far outside the range
    w = World(device, oracle, 2048, seed=13, termination_enabled=0)
    S = w.state.numpy()
    S[::3, 0] = 2.0e5
    S[1::3, 9] = -8.0e4
    w.state.set(S)
    w.policy.set_precision("f16x2")
    w.policy.reset()
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 30, "fused", autoreset=False)
    assert np.isfinite(w.policy.hidden_state(w.n)).all()
    assert np.isfinite(w.state.numpy()[:, 17:21]).all()      # the commands the policy issued


def test_sample_and_squash_layer(device, oracle, weights):
    """The full SampleAndSquash output stage (mean / log-std split + Philox sampling; not in the shipped checkpoint,
    semantics unpinned): evaluate_step against the oracle's restatement of the same definition, fused rollout ==
    chained rollout bit for bit in sampling mode, and the sampled spread where it can be predicted."""
    from raptor_amd.foundation_policy import Raptor
    rng = np.random.default_rng(3)
    B = 1000
    w_ls = (rng.standard_normal((4, 16)) * 0.3).astype(np.float32)
    b_ls = np.array([-1.0, -0.5, 0.2, -2.0], np.float32)
    pol = Raptor(device)
    pol.set_sample_and_squash("sample", w_ls, b_ls, seed=99)
    pol.reset()
    H = np.zeros((B, 16), np.float32)
    for step in range(3):
        obs = rng.standard_normal((B, 22)).astype(np.float32)
        got = pol.evaluate_step(obs)
        ref = oracle.actor_batch_step_sas(weights, w_ls, b_ls, 2, 99, step, 0, obs, H)
        assert np.abs(got - ref).max() < 2e-4, (step, np.abs(got - ref).max())      # Box-Muller on hardware transcendentals
        assert np.abs(got).max() <= 1.0
    pol.set_sample_and_squash("mean")
    pol.reset()
    H[:] = 0
    obs = rng.standard_normal((B, 22)).astype(np.float32)
    assert np.abs(pol.evaluate_step(obs) - oracle.actor_batch_step_sas(weights, None, None, 1, 0, 0, 0, obs, H)).max() < ACTOR_TOL
    # a state-independent log-std: the pre-squash sample is mean + sigma eps; with sigma = 0.05 the spread of
    # atanh(sample) - atanh(mean action) over many envs is sigma
    det, smp = Raptor(device), Raptor(device)
    det.set_sample_and_squash("mean")
    smp.set_sample_and_squash("sample", None, np.full(4, np.log(0.05), np.float32), seed=5)
    obs = (rng.standard_normal((20000, 22)) * 0.3).astype(np.float32)
    det.reset(); smp.reset()
    d = np.arctanh(np.clip(smp.evaluate_step(obs), -0.999999, 0.999999)) - np.arctanh(np.clip(det.evaluate_step(obs), -0.999999, 0.999999))
    keep = np.abs(det.evaluate_step(obs) if False else d) < 1.0
    assert abs(d[keep].std() - 0.05) < 0.003 and abs(d[keep].mean()) < 0.002
    # rollouts: fused and chained draw the same noise (counter = rng epoch, key = global env id)
    kw = dict(seed=17, episode_step_limit=40)
    a, b = World(device, oracle, 300, **kw), World(device, oracle, 300, **kw)
    for w_ in (a, b):
        w_.policy.set_sample_and_squash("sample", w_ls, b_ls, seed=7)
    ta, tb = a.vector.Trajectory(a.env, 60), b.vector.Trajectory(b.env, 60)
    a.vector.rollout(device, a.env, a.params, a.state, a.policy, a.rng, 60, "fused", True, trajectory=ta)
    b.vector.rollout(device, b.env, b.params, b.state, b.policy, b.rng, 60, "chained", True, trajectory=tb)
    A, Bt = ta.numpy(), tb.numpy()
    for k in ("obs", "act", "rew", "done"):
        assert np.array_equal(A[k], Bt[k]), k
    assert np.abs(A["act"]).max() <= 1.0 and A["act"].std() > 0.05
    with pytest.raises(Exception):
        a.policy.evaluate_sequence(np.zeros((3, 300, 22), np.float32))       # deterministic passes reject sampling


def test_the_readmes_own_simulator_snippet_as_written(device, oracle, weights):
    """README.md:17-25: `observation = np.array([[*sim.position, *R(sim.orientation).flatten(), ...]])` is a float64 array of one row,
    `policy.evaluate_step(observation)[0]` the action; Raptor() takes no device.  Whatever dtype or container the caller builds -
    float64 array, nested list, float32 array - the policy sees the same float32 rows and answers the same bits, call after call
    (the third call on goes to the resident policy executor; float64 / list inputs take the converting path in front of it)."""
    from raptor_amd.foundation_policy import Raptor
    rng = np.random.default_rng(3)
    rows = [rng.standard_normal(22) for _ in range(60)]
    answers = []
    for kind in ("float64", "list", "float32"):
        policy = Raptor()                                  # README.md:20: no arguments
        policy.reset()
        out = []
        for r in rows:
            position, rot, vel, omega, action = r[0:3], r[3:12].reshape(3, 3), r[12:15], r[15:18], r[18:22]
            observation = np.array([[*position, *rot.flatten(), *vel, *omega, *action]])          # README.md:23 (float64)
            assert observation.dtype == np.float64 and observation.shape == (1, 22)
            if kind == "list":
                observation = observation.tolist()
            elif kind == "float32":
                observation = observation.astype(np.float32)
            a = policy.evaluate_step(observation)[0]       # README.md:24
            assert a.shape == (4,) and a.dtype == np.float32
            out.append(a.copy())
        answers.append(np.array(out))
    assert np.array_equal(answers[0].view(np.uint32), answers[1].view(np.uint32))
    assert np.array_equal(answers[0].view(np.uint32), answers[2].view(np.uint32))
    H = np.tile(weights[2000:2016], (1, 1)).astype(np.float32)
    for t, r in enumerate(rows):
        ref = oracle.actor_batch_step(weights, r.astype(np.float32)[None, :], H)
        assert np.max(np.abs(answers[0][t] - ref[0])) < 1e-5, t
