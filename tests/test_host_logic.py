import numpy as np
import pytest

from raptor_amd.distributed import shard_range, shard_sizes


@pytest.mark.parametrize("n,w", [(8, 1), (8, 2), (65536, 8), (2097152, 8), (10, 3), (7, 8), (1, 4)])
def test_shard_range_partitions_exactly(n, w):
    covered = []
    for r in range(w):
        s, c = shard_range(n, w, r)
        covered.extend(range(s, s + c))
    assert covered == list(range(n))
    sizes = shard_sizes(n, w)
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n


def test_shard_range_rejects_bad_rank():
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_policy_weights_file():
    from raptor_amd.foundation_policy import load_weights
    w = load_weights()
    assert w.dtype == np.float32 and w.size == 2084 and np.isfinite(w).all()


def test_checkpoint_header_round_trip(tmp_path):
    """The reference's C++ export format (byte-array blobs per namespace) written and parsed back."""
    from raptor_amd.checkpoint import load_checkpoint_header, write_checkpoint_header
    from raptor_amd.foundation_policy import load_weights
    w = load_weights()
    x = np.random.default_rng(0).standard_normal((5, 2, 22)).astype(np.float32)
    y = np.random.default_rng(1).standard_normal((5, 2, 4)).astype(np.float32)
    path = tmp_path / "checkpoint.h"
    write_checkpoint_header(path, w, (x, y))
    w2, ex = load_checkpoint_header(path)
    assert np.array_equal(w, w2)
    assert ex is not None and np.array_equal(ex[0], x) and np.array_equal(ex[1], y)
    bad = tmp_path / "bad.h"
    bad.write_text("namespace rl_tools::checkpoint::actor {\n}\n")
    with pytest.raises(ValueError):
        load_checkpoint_header(bad)


def test_checkpoint_header_of_the_reference(tmp_path):
    """When the reference tarball is present (build container only), its checkpoint.h parses to the
    shipped weights and carries the first known-answer vector."""
    import os
    import tarfile
    tar = "/root/reference/data/raptor-policy-checkpoint.tar.gz"
    if not os.path.exists(tar):
        pytest.skip("reference not present on this box")
    from raptor_amd.checkpoint import load_checkpoint_header
    from raptor_amd.foundation_policy import load_weights
    from conftest import GOLDEN
    with tarfile.open(tar) as tf:
        tf.extract("2025-04-19_16-16-17/checkpoint.h", tmp_path)
    w, ex = load_checkpoint_header(tmp_path / "2025-04-19_16-16-17" / "checkpoint.h")
    assert np.array_equal(w, load_weights())
    assert ex[0].shape == (500, 2, 22)
    assert np.array_equal(ex[0].ravel(), np.fromfile(os.path.join(GOLDEN, "kat_h_input.bin"), "<f4"))
    assert np.array_equal(ex[1].ravel(), np.fromfile(os.path.join(GOLDEN, "kat_h_output.bin"), "<f4"))


def test_hdf5_checkpoint_reader(oracle):
    """The published checkpoint.h5 (fixture) through the dependency-free HDF5 reader: same weights
    as the shipped parameter file, the second known-answer pair, the observation spec string, and
    the oracle reproduces that pair from these weights."""
    import os
    from conftest import GOLDEN
    from raptor_amd.checkpoint import load_checkpoint, load_checkpoint_h5
    from raptor_amd.foundation_policy import load_weights
    path = os.path.join(GOLDEN, "checkpoint.h5")
    w, ex, meta = load_checkpoint_h5(path)
    assert np.array_equal(w, load_weights())
    assert "OrientationRotationMatrix" in meta and "ActionHistory(1)" in meta
    assert np.array_equal(ex[0].ravel(), np.fromfile(os.path.join(GOLDEN, "kat_h5_input.bin"), "<f4"))
    assert np.array_equal(ex[1].ravel(), np.fromfile(os.path.join(GOLDEN, "kat_h5_output.bin"), "<f4"))
    assert np.abs(oracle.actor_sequence(w, ex[0]) - ex[1]).max() < 1e-5
    w2, ex2 = load_checkpoint(path)
    assert np.array_equal(w2, w) and ex2[0].shape == (500, 2, 22)


def test_hdf5_reader_rejects_garbage(tmp_path):
    from raptor_amd.hdf5_min import File, Hdf5FormatError
    p = tmp_path / "x.h5"
    p.write_bytes(b"not hdf5 at all")
    with pytest.raises(Hdf5FormatError):
        File(p)
