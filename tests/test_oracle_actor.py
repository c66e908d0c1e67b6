"""The oracle's actor restatement is pinned by the reference's own known-answer vectors."""
import hashlib
import json
import os

import numpy as np

from conftest import GOLDEN, ROOT

TOL = 1e-5   # fp32 abs tolerance on raw actions in [-2.8, 3.4] (achievable 7e-7)


def test_golden_manifest_intact():
    man = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))
    for rel, meta in man.items():
        if rel.startswith("_"):
            continue
        data = open(os.path.join(ROOT, rel), "rb").read()
        assert hashlib.sha256(data).hexdigest() == meta["sha256"], rel
        assert len(data) == (1 if meta["dtype"] == "bytes" else 4) * int(np.prod(meta["shape"]))


def test_kat_vectors_are_independent():
    a = np.fromfile(os.path.join(GOLDEN, "kat_h_input.bin"), "<f4")
    b = np.fromfile(os.path.join(GOLDEN, "kat_h5_input.bin"), "<f4")
    assert not np.array_equal(a, b)


def test_actor_sequence_matches_kat(oracle, weights, kat):
    x, y = kat
    out = oracle.actor_sequence(weights, x)
    err = np.abs(out - y).max()
    assert err < TOL, err


def test_actor_step_equals_sequence(oracle, weights, kat):
    """reset(); 500 x evaluate_step == the sequence evaluation (same hidden-state recurrence)."""
    x, _ = kat
    seq = oracle.actor_sequence(weights, x)
    h = np.tile(weights[2000:2016], (2, 1)).astype(np.float32)
    for t in range(x.shape[0]):
        a = oracle.actor_batch_step(weights, x[t], h)
        assert np.array_equal(a, seq[t])


def test_actor_is_recurrent(oracle, weights, kat):
    """Resetting the hidden state every step must NOT reproduce the KAT (dim 0 is time)."""
    x, y = kat
    out = np.stack([oracle.actor_sequence(weights, x[t:t + 1])[0] for t in range(50)])
    assert np.abs(out - y[:50]).max() > 0.1


def test_actor_wider_observation_stride(oracle, weights, kat):
    """Only the first 22 columns are read (caller slices [:, :22], README.md:97)."""
    x, _ = kat
    wide = np.concatenate([x[0], np.full((2, 4), 123.0, np.float32)], axis=1)
    h1 = np.zeros((2, 16), np.float32)
    h2 = np.zeros((2, 16), np.float32)
    assert np.array_equal(oracle.actor_batch_step(weights, x[0], h1), oracle.actor_batch_step(weights, wide, h2))


def test_initial_hidden_state_is_zero(weights):
    assert np.all(weights[2000:2016] == 0.0)   # checkpoint.h:123
