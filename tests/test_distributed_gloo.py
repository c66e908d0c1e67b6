"""N>1 path on CPU: two gloo ranks each own a contiguous shard of one logical batch (RNG keyed
by GLOBAL env id), roll it out (with the oracle, since there is no GPU here), all-gather the
episode returns with raptor_amd.distributed and must reproduce the single-process result
bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

N_TOTAL = 101   # odd on purpose: uneven shards
STEPS = 60


def _rollout_shard(start, count):
    from oracle import oracle as O
    w = np.fromfile(os.path.join(ROOT, "raptor_amd", "data", "raptor_policy.bin"), "<f4")
    cfg = O.default_config()
    cfg.episode_step_limit = 25
    P = O.sample_initial_parameters(cfg, 5, 0, start, count)
    st = O.Stats(count)
    S = O.sample_initial_state(cfg, 5, st.episode, start, P)
    H = np.zeros((count, 16), np.float32)
    O.rollout(cfg, w, 5, 0, start, P, S, H, STEPS, 1, st)
    return st.fin_returns.copy()


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raptor_amd.distributed import all_gather_returns, shard_range
    try:
        start, count = shard_range(N_TOTAL, world, rank)
        local = torch.from_numpy(_rollout_shard(start, count))
        full = all_gather_returns(local, N_TOTAL)
        # even shards exercise all_gather_into_tensor
        even = all_gather_returns(torch.full((4,), float(rank)), 4 * world)
        q.put((rank, full.numpy().copy(), even.numpy().copy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_two_rank_sharded_rollout_equals_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single = _rollout_shard(0, N_TOTAL)
    for rank, full, even in results:
        assert full.shape == (N_TOTAL,)
        assert np.array_equal(full, single), f"rank {rank}: sharded result differs from single-process"
        assert even.tolist() == [0.0] * 4 + [1.0] * 4


def _exchange_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raptor_amd.distributed import ReturnsExchange
    try:
        n = 6
        ex = ReturnsExchange(n, n * world, "cpu")
        seen = []
        for episode in range(5):               # 5 posts over 2 buffers: every buffer is reused at least once
            ex.post(lambda buf, e=episode: buf.copy_(torch.arange(n, dtype=torch.float32) + 100 * rank + 1000 * e))
            if episode in (1, 4):
                seen.append(ex.finish().clone().numpy())
        q.put((rank, seen))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_overlapped_returns_exchange():
    """ReturnsExchange (the per-episode all-gather bench.py overlaps with the next rollout): every rank
    sees every rank's buffer of the LAST post, also after the double buffers were recycled."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    base = np.arange(6, dtype=np.float32)
    for rank, seen in results:
        for got, episode in zip(seen, (1, 4)):
            want = np.concatenate([base + 100 * r + 1000 * episode for r in range(world)])
            assert np.array_equal(got, want), (rank, episode, got)


def test_returns_exchange_single_process():
    from raptor_amd.distributed import ReturnsExchange
    ex = ReturnsExchange(4, 4, "cpu")
    assert ex.finish() is None
    for e in range(3):
        ex.post(lambda buf, e=e: buf.fill_(float(e)))
    assert ex.finish().tolist() == [2.0] * 4
    with pytest.raises(ValueError):
        ReturnsExchange(4, 5, "cpu")
