"""the teacher bank (DAgger relabel, README.md:208-216) against the oracle.

Parity of the HIP path (through the C ABI of libraptor_quad.so) against the oracle.  Bars (DESIGN.md "Parity"):
  * integer / index / mask work, parameter sampling, observe (no noise) and env transitions for identical inputs: BIT-EXACT;
  * anything behind a transcendental (actor gates, sin/cos of the initial attitude, Box-Muller noise): float32 tolerance stated per test;
  * the actor additionally against the reference's own known-answer vectors (< 1e-5).
(Round 6 split tests/test_gpu_parity.py - 2 987 lines, one module - by SURVEY.md section 8 row group, so that a red run names its row.)
"""
import os

import numpy as np
import pytest

from gpu_common import World      # noqa: F401

pytestmark = pytest.mark.gpu

# ------------------------------------------------------------------------------ teacher bank -
def _teacher_weights(rng, n_teachers, in_dim, h1, h2):
    from raptor_amd.teachers import parameter_count
    W = np.empty((n_teachers, parameter_count(in_dim, h1, h2)), np.float32)
    for t in range(n_teachers):      # He-style scales per layer so that activations stay O(1) through the net
        parts = [rng.standard_normal(h1 * in_dim) / np.sqrt(in_dim), rng.standard_normal(h1) * 0.1,
                 rng.standard_normal(h2 * h1) / np.sqrt(h1), rng.standard_normal(h2) * 0.1,
                 rng.standard_normal(4 * h2) / np.sqrt(h2), rng.standard_normal(4) * 0.1]
        W[t] = np.concatenate(parts).astype(np.float32)
    return W


ACT_CODE = {"identity": 0, "relu": 1, "tanh": 2}


@pytest.mark.parametrize("h1,h2,act,out_act,in_dim", [(64, 64, "relu", "identity", 22), (64, 64, "tanh", "tanh", 22),
                                                      (32, 16, "relu", "tanh", 18), (16, 64, "tanh", "identity", 22),
                                                      (64, 32, "relu", "identity", 13)])
def test_teacher_bank_relabel_vs_oracle(device, oracle, h1, h2, act, out_act, in_dim):
    """MLP teachers on a recorded trajectory: f32 MFMA path within 1e-5 of the oracle's fma chains, bf16 path
    within 5e-2; ragged teacher groups (sizes 1..50, not multiples of the 16-env tile), interleaved ids."""
    from raptor_amd.teachers import TeacherBank
    rng = np.random.default_rng(h1 * 1000 + h2)
    n, T, n_teachers = 1000, 6, 37
    w = World(device, oracle, n, seed=31, episode_step_limit=4)
    tr = w.vector.Trajectory(w.env, T)
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, T, "fused", True, trajectory=tr)
    rec = tr.numpy()
    W = _teacher_weights(rng, n_teachers, in_dim, h1, h2)
    ids = rng.integers(0, n_teachers, n).astype(np.uint32)
    ids[:100] = np.arange(100) % 5                      # interleaved
    bank = TeacherBank(device, W, in_dim, h1, h2, act, out_act)
    ref = oracle.teacher_relabel(W, in_dim, h1, h2, ACT_CODE[act], ACT_CODE[out_act], rec["obs"], ids, 4)
    got = tr.relabel_teachers(bank, ids)
    assert np.abs(got - ref).max() < 1e-5, np.abs(got - ref).max()
    assert np.array_equal(tr.numpy()["act"], rec["act"])          # overwrite=False leaves the recording alone
    bank.set_precision("bf16")
    got16 = tr.relabel_teachers(bank, ids)
    assert np.abs(got16 - ref).max() < 5e-2, np.abs(got16 - ref).max()
    bank.set_precision("f16x2")                                     # two f16 pieces per operand: the fp32 bar
    got_split = tr.relabel_teachers(bank, ids)
    assert np.abs(got_split - ref).max() < 1e-5, np.abs(got_split - ref).max()
    bank.set_precision("fp32")
    tr.relabel_teachers(bank, ids, overwrite=True, fetch=False)
    assert np.array_equal(tr.numpy()["act"], got)                 # overwrite=True: the stored actions are the labels
    with pytest.raises(Exception):
        tr.relabel_teachers(bank, np.full(n, n_teachers, np.uint32))      # id out of range


def _stack_weights(rng, n_teachers, in_dim, widths, scale=0.3):
    dims = [in_dim] + list(widths) + [4]
    per = sum(dims[i + 1] * dims[i] + dims[i + 1] for i in range(len(dims) - 1))
    return (rng.standard_normal((n_teachers, per)) * scale / np.sqrt(max(widths) / 16.0)).astype(np.float32)


@pytest.mark.parametrize("in_dim,widths,act,out_act", [(22, [128, 128, 128], "relu", "identity"), (22, [128, 48, 112], "tanh", "tanh"),
                                                       (22, [96], "relu", "tanh"), (13, [32, 16, 64], "tanh", "identity"),
                                                       (22, [64, 128], "relu", "identity"), (9, [16], "tanh", "identity")])
def test_teacher_bank_dense_stacks_vs_oracle(device, oracle, in_dim, widths, act, out_act):
    """Round 5: teachers outside the register-stationary family - one or three hidden layers, widths up to 128, ragged widths
    (padded to 64 / 128 units with exact zeros) - through the streaming fp32 kernel k_teacher_relabel_layers, within 1e-5 of the
    oracle's fma chains; ragged teacher groups, interleaved ids; bf16 / split-f16 are refused for such a bank."""
    from raptor_amd.teachers import TeacherBank, layers_parameter_count
    rng = np.random.default_rng(sum(widths) * 7 + in_dim)
    n, T, n_teachers = 777, 5, 23
    w = World(device, oracle, n, seed=33, episode_step_limit=4)
    tr = w.vector.Trajectory(w.env, T)
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, T, "fused", True, trajectory=tr)
    rec = tr.numpy()
    W = _stack_weights(rng, n_teachers, in_dim, widths)
    assert W.shape[1] == layers_parameter_count(in_dim, widths)
    ids = rng.integers(0, n_teachers, n).astype(np.uint32)
    ids[:60] = np.arange(60) % 3
    bank = TeacherBank.from_layers(device, W, in_dim, widths, act, out_act)
    ref = oracle.mlp_relabel(W, in_dim, widths, ACT_CODE[act], ACT_CODE[out_act], rec["obs"], ids, 4)
    got = tr.relabel_teachers(bank, ids)
    assert np.abs(got - ref).max() < 1e-5, np.abs(got - ref).max()
    assert np.array_equal(got, tr.relabel_teachers(bank, ids))              # and the same bits twice
    for prec in ("bf16", "f16x2"):
        with pytest.raises(Exception, match="fp32 only"):
            bank.set_precision(prec)
    with pytest.raises(ValueError):
        TeacherBank.from_layers(device, W, in_dim, widths + [16, 16] if len(widths) > 1 else [24], act, out_act)     # four layers / a width of 24


def test_teacher_bank_from_a_thousand_checkpoint_files(device, oracle, tmp_path):
    """The row most likely to meet real data (round 4's verdict): 1 000 teachers, one HDF5 file each in the reference's layout
    (`sequential` of `dense` layers, h5:/actor/layers/*; written by this package's own writer), loaded by
    TeacherBank.from_checkpoints and evaluated on a recorded trajectory: identical to the bank built from the same arrays, and
    within 1e-5 of the oracle.  Then a three-hidden-layer, 128-wide set the same way (the streaming kernel)."""
    from raptor_amd.checkpoint import write_mlp_checkpoint_h5
    from raptor_amd.teachers import TeacherBank, balanced_teacher_assignment
    rng = np.random.default_rng(99)
    n, T = 16000, 4
    w = World(device, oracle, n, seed=34, episode_step_limit=4)
    tr = w.vector.Trajectory(w.env, T)
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, T, "fused", True, trajectory=tr)
    obs = tr.numpy()["obs"]
    for n_teachers, widths, act, tag in ((1000, [64, 64], "relu", "a"), (12, [128, 128, 128], "tanh", "b")):
        dims = [22] + widths + [4]
        paths, blocks = [], []
        for k in range(n_teachers):
            layers = [((rng.standard_normal((dims[i + 1], dims[i])) * 0.3 / np.sqrt(dims[i] / 16.0)).astype(np.float32),
                       (rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32)) for i in range(len(dims) - 1)]
            path = str(tmp_path / f"teacher_{tag}_{k}.h5")
            write_mlp_checkpoint_h5(path, layers, [act] * len(widths) + ["identity"])
            paths.append(path)
            blocks.append(np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for W, b in layers]))
        bank = TeacherBank.from_checkpoints(device, paths)
        assert bank.n_teachers == n_teachers and bank.widths == widths and bank.hidden_activation == act
        ids = balanced_teacher_assignment(n, n_teachers)
        got = tr.relabel_teachers(bank, ids)
        ref = oracle.mlp_relabel(np.stack(blocks), 22, widths, ACT_CODE[act], 0, obs, ids, 8)
        assert np.abs(got - ref).max() < 1e-5, (widths, np.abs(got - ref).max())
        direct = TeacherBank.from_layers(device, np.stack(blocks), 22, widths, act, "identity")
        assert np.array_equal(got, tr.relabel_teachers(direct, ids))
    # files that do not agree on the topology are refused, naming the file
    odd = str(tmp_path / "odd.h5")
    write_mlp_checkpoint_h5(odd, [(np.zeros((32, 22), np.float32), np.zeros(32, np.float32)), (np.zeros((4, 32), np.float32), np.zeros(4, np.float32))],
                            ["relu", "identity"])
    with pytest.raises(ValueError, match="odd.h5"):
        TeacherBank.from_checkpoints(device, [paths[0], odd])


def test_teacher_bank_at_full_batch(device, oracle):
    """65 536 envs x 64 teachers: f32 path against the oracle on every env."""
    from raptor_amd.teachers import TeacherBank
    rng = np.random.default_rng(77)
    n, T, n_teachers = 65536, 4, 64
    w = World(device, oracle, n, seed=32)
    tr = w.vector.Trajectory(w.env, T)
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, T, "fused", True, trajectory=tr)
    obs = tr.numpy()["obs"]
    W = _teacher_weights(rng, n_teachers, 22, 64, 64)
    ids = (np.arange(n) // 1024).astype(np.uint32)               # 1024 envs per teacher, as a learner would shard them
    rng.shuffle(ids[:4096])                                       # and a shuffled corner
    bank = TeacherBank(device, W, 22, 64, 64, "relu", "identity")
    ref = oracle.teacher_relabel(W, 22, 64, 64, 1, 0, obs, ids, 8)
    got = tr.relabel_teachers(bank, ids)
    assert np.abs(got - ref).max() < 1e-5, np.abs(got - ref).max()
    bank.set_precision("bf16")
    assert np.abs(tr.relabel_teachers(bank, ids) - ref).max() < 5e-2
    bank.set_precision("f16x2")
    err = np.abs(tr.relabel_teachers(bank, ids) - ref).max()
    print(f"\n[teacher bank, 65 536 envs x 64 teachers] max |label - oracle|: f32 MFMA {np.abs(got - ref).max():.2e}, split f16 {err:.2e}")
    assert err < 1e-5, err


def test_teacher_bank_edge_cases(device, oracle):
    """One env, one step, one teacher; an empty trajectory; a bank whose teachers nobody uses; bad arguments."""
    from raptor_amd.teachers import TeacherBank, parameter_count
    rng = np.random.default_rng(8)
    w = World(device, oracle, 1, seed=51)
    tr = w.vector.Trajectory(w.env, 2)
    W = _teacher_weights(rng, 3, 22, 16, 16)
    bank = TeacherBank(device, W, 22, 16, 16, "tanh", "identity")
    out = tr.relabel_teachers(bank, np.array([2], np.uint32))            # nothing recorded yet
    assert out.shape == (0, 1, 4)
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 1, "fused", True, trajectory=tr)
    got = tr.relabel_teachers(bank, np.array([2], np.uint32))
    ref = oracle.teacher_relabel(W, 22, 16, 16, 2, 0, tr.numpy()["obs"], np.array([2], np.uint32))
    assert got.shape == (1, 1, 4) and np.abs(got - ref).max() < 1e-5
    for bad in (dict(in_dim=23), dict(h1=48), dict(hidden_activation="identity")):
        kw = dict(in_dim=22, h1=16, h2=16, hidden_activation="relu", output_activation="identity")
        kw.update(bad)
        with pytest.raises(Exception):
            TeacherBank(device, np.zeros((1, parameter_count(kw["in_dim"], kw["h1"], kw["h2"])), np.float32), **kw)
    with pytest.raises(ValueError):
        TeacherBank(device, np.zeros((2, 7), np.float32))                 # wrong parameter count
