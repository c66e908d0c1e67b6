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
