"""auto-reset + trajectory writer [T, N, {obs22, act4, rew, done}]; relabelling a recorded trajectory with a policy.

Parity of the HIP path (through the C ABI of libraptor_quad.so) against the oracle.  Bars (DESIGN.md "Parity"):
  * integer / index / mask work, parameter sampling, observe (no noise) and env transitions for identical inputs: BIT-EXACT;
  * anything behind a transcendental (actor gates, sin/cos of the initial attitude, Box-Muller noise): float32 tolerance stated per test;
  * the actor additionally against the reference's own known-answer vectors (< 1e-5).
(Round 6 split tests/test_gpu_parity.py - 2 987 lines, one module - by SURVEY.md section 8 row group, so that a red run names its row.)
"""
import os

import numpy as np
import pytest

from gpu_common import ACTOR_TOL, INIT_TOL, World      # noqa: F401

pytestmark = pytest.mark.gpu

# ------------------------------------------------------------------------------ trajectory -
@pytest.mark.parametrize("autoreset", [False, True])
def test_trajectory_fused_equals_chained(device, oracle, autoreset):
    kw = dict(seed=14, episode_step_limit=30, noise_position=0.01)
    a, b = World(device, oracle, 200, **kw), World(device, oracle, 200, **kw)
    ta, tb = a.vector.Trajectory(a.env, 80), b.vector.Trajectory(b.env, 80)
    for chunk in (50, 30):
        a.vector.rollout(device, a.env, a.params, a.state, a.policy, a.rng, chunk, "fused", autoreset, trajectory=ta)
        b.vector.rollout(device, b.env, b.params, b.state, b.policy, b.rng, chunk, "chained", autoreset, trajectory=tb)
    A, B = ta.numpy(), tb.numpy()
    assert len(ta) == len(tb) == 80 and A["obs"].shape == (80, 200, 22)
    assert np.array_equal(A["done"], B["done"])
    live = A["done"] != 4
    for k in ("obs", "act", "rew"):
        assert np.array_equal(A[k][live], B[k][live]), k
    if autoreset:
        assert live.all() and (A["done"] == 2).sum() >= 2 * 200 - (A["done"] == 1).sum() * 2 - 200
    else:
        assert (A["done"][40:] == 4).all()          # every env ended by step 30 and froze
        assert ((A["done"] == 1) | (A["done"] == 2)).sum() == 200
    with pytest.raises(Exception):                   # capacity exhausted
        a.vector.rollout(device, a.env, a.params, a.state, a.policy, a.rng, 1, "fused", autoreset, trajectory=ta)


def test_autoreset_after_a_freezing_rollout_thaws_frozen_envs(device, oracle, weights):
    """A rollout WITHOUT auto-reset leaves envs frozen; a later rollout WITH auto-reset must start their next
    episode (re-sampled state, policy state reset) before its first step - in the fused kernel's prologue and in
    the chained mode's thaw launch alike - and record real transitions for them (never done code 4).  Both modes
    bit for bit, and against the oracle's recorded rollout of the same history."""
    kw = dict(seed=21, episode_step_limit=25, noise_position=0.01)
    a, b = World(device, oracle, 300, **kw), World(device, oracle, 300, **kw)
    a.sync_oracle_to_gpu_state()
    for w, mode in ((a, "fused"), (b, "chained")):
        w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 40, mode, False)
        assert w.env.frozen().all()                  # every episode ended within 25 steps
    ta, tb = a.vector.Trajectory(a.env, 30), b.vector.Trajectory(b.env, 30)
    a.vector.rollout(device, a.env, a.params, a.state, a.policy, a.rng, 30, "fused", True, trajectory=ta)
    b.vector.rollout(device, b.env, b.params, b.state, b.policy, b.rng, 30, "chained", True, trajectory=tb)
    A, B = ta.numpy(), tb.numpy()
    assert (A["done"] != 4).all() and np.array_equal(A["done"], B["done"])
    for k in ("obs", "act", "rew"):
        assert np.array_equal(A[k], B[k]), k
    assert np.array_equal(a.state.numpy(), b.state.numpy())
    assert np.array_equal(a.policy.hidden_state(300), b.policy.hidden_state(300))
    assert not a.env.frozen().any() and not b.env.frozen().any()
    assert np.array_equal(a.env.episode_index(), b.env.episode_index()) and (a.env.episode_index() >= 2).all()
    # the oracle through the same history: its first recorded observation is the thawed (re-sampled) state
    oracle.rollout(a.cfg, weights, 21, 0, 0, a.P, a.S, a.H, 40, 0, a.st, 4)
    assert a.st.frozen.all()
    ref = oracle.rollout_record(a.cfg, weights, 21, 40, 0, a.P, a.S, a.H, 30, 1, a.st, 4)
    assert np.array_equal(ref["done"][0], A["done"][0])
    assert np.abs(A["obs"][0][:, :3] - ref["obs"][0][:, :3]).max() <= 0.011 * 6      # position + N(0, 0.01) noise
    assert np.array_equal(a.env.episode_index(), a.st.episode)


def test_trajectory_vs_oracle(device, oracle, weights):
    """Recorded transitions against the oracle's, teacher-forced start, 3 steps, auto-reset with a
    2-step episode limit so that the reset path is inside the window."""
    w = World(device, oracle, 512, seed=15, episode_step_limit=2)
    w.sync_oracle_to_gpu_state()
    tr = w.vector.Trajectory(w.env, 3)
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 3, "fused", True, trajectory=tr)
    ref = oracle.rollout_record(w.cfg, weights, 15, 0, 0, w.P, w.S, w.H, 3, 1, w.st, 4)
    G = tr.numpy()
    assert np.array_equal(G["done"], ref["done"]) and (G["done"][1] >= 1).all()
    assert np.array_equal(G["obs"][0], ref["obs"][0])                 # same state in -> same bits out
    assert np.abs(G["act"][0] - ref["act"][0]).max() < ACTOR_TOL
    assert np.abs(G["rew"] - ref["rew"]).max() < 1e-4
    assert np.abs(G["obs"][1] - ref["obs"][1]).max() < 1e-4
    # step 2 observes the freshly re-sampled state (episode 2): independent of the actor
    assert np.abs(G["obs"][2][:, :3] - ref["obs"][2][:, :3]).max() == 0.0
    assert np.abs(G["obs"][2] - ref["obs"][2]).max() < INIT_TOL * 4
    assert np.abs(w.state.numpy()[:, :13] - w.S[:, :13]).max() < 1e-4


@pytest.mark.parametrize("autoreset", [False, True])
@pytest.mark.parametrize("n", [300, 70000])
def test_trajectory_relabel_with_the_recording_policy_is_the_identity(device, oracle, n, autoreset):
    """Relabelling a recorded rollout with the policy that produced it must give back the stored actions bit
    for bit on every step that was taken - through episode ends (GRU reset), auto-resets and past frozen envs."""
    from raptor_amd.foundation_policy import Raptor
    w = World(device, oracle, n, seed=31, episode_step_limit=9, termination_position=0.6)
    T = 64
    traj = w.vector.Trajectory(w.env, T)
    w.policy.reset()
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, T, "fused", autoreset=autoreset, trajectory=traj)
    rec = traj.numpy()
    assert {0, 2}.issubset(set(np.unique(rec["done"]).tolist()))          # episodes did end inside the recording
    if not autoreset:
        assert 4 in np.unique(rec["done"])                                 # and envs froze
    teacher = Raptor(device)                      # same weights, separate object and hidden state
    teacher.reset()
    relabelled = traj.relabel(teacher)
    live = rec["done"] != 4                       # steps of frozen envs carry no defined action in a recording
    assert live.sum() >= 9 * n // 2 and np.array_equal(relabelled[live], rec["act"][live])
    assert np.array_equal(traj.numpy()["act"], rec["act"])                 # overwrite=False left the buffer alone


def test_trajectory_relabel_across_a_frozen_stretch(device, oracle):
    """One recording made of two rollouts: the first without auto-reset (envs freeze when their episode ends, code 4 for
    the rest of it), the second with (the frozen envs thaw: new episode, policy state reset).  The relabel kernel carries
    recurrent accumulators from one step into the next (round 3): a frozen step must leave them as they were, an episode
    end must replace them - the recording policy has to get its own actions back bit for bit on every live step, the
    first one after the frozen stretch included."""
    from raptor_amd.foundation_policy import Raptor
    w = World(device, oracle, 1000, seed=35, episode_step_limit=13, termination_position=0.7)
    T1, T2 = 30, 30
    traj = w.vector.Trajectory(w.env, T1 + T2)
    w.policy.reset()
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, T1, "fused", autoreset=False, trajectory=traj)
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, T2, "fused", autoreset=True, trajectory=traj)
    rec = traj.numpy()
    assert (rec["done"][:T1] == 4).any() and not (rec["done"][T1:] == 4).any()
    thawed = rec["done"][T1 - 1] == 4                 # frozen at the end of the first rollout, stepping again in the second
    assert thawed.sum() > 100
    teacher = Raptor(device)
    teacher.reset()
    relabelled = traj.relabel(teacher)
    live = rec["done"] != 4
    assert np.array_equal(relabelled[live], rec["act"][live])
    assert np.array_equal(relabelled[T1][thawed], rec["act"][T1][thawed])      # the first step after the frozen stretch


def test_trajectory_relabel_with_another_policy(device, oracle, weights):
    """A different policy (perturbed weights, standing in for a teacher) on the recorded observations: equals
    the oracle's actor run over each env's observation sequence with a reset after every recorded episode end;
    overwrite=True replaces the stored actions."""
    from raptor_amd.foundation_policy import Raptor
    n, T = 200, 40
    w = World(device, oracle, n, seed=33, episode_step_limit=11)
    traj = w.vector.Trajectory(w.env, T)
    w.policy.reset()
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, T, "fused", autoreset=True, trajectory=traj)
    rec = traj.numpy()
    w2 = (weights + np.random.default_rng(0).standard_normal(weights.size).astype(np.float32) * 0.02).astype(np.float32)
    w2[2000:2016] = 0.05                          # a non-zero initial hidden state makes the resets visible
    teacher = Raptor(device, weights=w2)
    teacher.reset()
    got = traj.relabel(teacher, overwrite=True)
    H = np.tile(w2[2000:2016], (n, 1)).astype(np.float32)
    for t in range(T):
        ref = oracle.actor_batch_step(w2, np.ascontiguousarray(rec["obs"][t]), H)
        assert np.max(np.abs(got[t] - ref)) < ACTOR_TOL, t
        ended = (rec["done"][t] == 1) | (rec["done"][t] == 2)
        H[ended] = w2[2000:2016]
    assert np.array_equal(traj.numpy()["act"], got)
    assert not np.array_equal(got, rec["act"])


def test_collect_and_relabel_example_runs(tmp_path):
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "collect_and_relabel.py"), "--envs", "2048",
                        "--steps", "40"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "episode ends" in r.stdout and "cuda" in r.stdout
