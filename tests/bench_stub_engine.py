"""A stand-in for the product behind bench.py's orchestration (bench.run_benchmark): the same interface as
bench.GpuEngine, no GPU.  TESTS ONLY - it exists so that the N > 1 path of bench.py (rendezvous, two-phase
consensus on the native communicator, fallback, per-episode exchange, max over ranks, the one JSON line) runs
with world_size 2 on the gloo backend before an 8-GPU node ever sees it.

    RANK=0 WORLD_SIZE=2 ... python bench.py --gpus 2 --engine tests.bench_stub_engine --backend gloo ...

Failure injection (environment): RQ_STUB_FAIL_PHASE1=<rank> makes that rank fail to produce a communicator id,
RQ_STUB_FAIL_PHASE2=<rank> makes its communicator creation fail - the other ranks must not be left waiting;
RQ_STUB_WRONG_RANKS=<k> makes the communicator describe itself as spanning k ranks, RQ_STUB_SCRAMBLE=1 makes the gathered
blocks come back rotated by one rank - either way bench.py must refuse to print a value."""
import os
import time

import numpy as np
import torch
import torch.distributed as dist


class _StubEnv:
    def __init__(self, n, offset):
        self.n, self.offset = n, offset
        self.returns = np.zeros(n, np.float32)
        self.episodes = 0

    def finished_returns(self, out=None, wait=True):
        if out is None:
            return self.returns.copy()
        out.copy_(torch.from_numpy(self.returns))
        return out


class _StubShard:
    """`rollout` takes ~1 us per step and, per finished 500-step episode, sets return[i] = global id + 1000 * episode."""

    def __init__(self, n, offset):
        self.n, self.env, self.steps = n, _StubEnv(n, offset), 0

    def rollout(self, steps, mode):
        t_end = time.perf_counter() + steps * 1e-6
        while time.perf_counter() < t_end:
            pass
        before = self.steps // 500
        self.steps += steps
        if self.steps // 500 != before:
            self.env.episodes = self.steps // 500
            self.env.returns[:] = np.arange(self.n, dtype=np.float32) + self.env.offset + 1000.0 * self.env.episodes


class _StubNativeExchange:
    """all_gather through the process group, shaped like raptor_amd.distributed.NativeReturnsExchange behind
    bench._NativeExchange (post / finish / result)."""
    kind = "native RCCL (rq_allgather_returns)"       # what the real adapter reports: the JSON field is asserted on

    def __init__(self, world):
        self.world, self.work, self.out = world, None, None
        self.ex = self                                  # bench.py tells native from torch-side by this attribute

    def post(self, shard):
        self.finish()
        local = torch.from_numpy(shard.env.returns.copy())
        self.out = torch.empty(self.world * local.numel(), dtype=torch.float32)
        self.work = dist.all_gather_into_tensor(self.out, local, async_op=True) if dist.is_initialized() else None
        if self.work is None:
            self.out = local

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.out

    def result(self):
        out = self.finish()
        if out is None:
            return None
        got = out.numpy()
        if os.environ.get("RQ_STUB_SCRAMBLE") and self.world > 1:      # a gather that lands the ranks' blocks in the wrong places
            got = np.roll(got.reshape(self.world, -1), 1, axis=0).reshape(-1).copy()
        return got

    def describe(self):
        rank = dist.get_rank() if dist.is_initialized() else 0
        fake = os.environ.get("RQ_STUB_WRONG_RANKS")          # a communicator that spans fewer ranks than the job
        return {"ranks": int(fake) if fake else self.world, "rank": rank, "version": "stub", "version_code": 0, "device": rank,
                "pci_bus_id": f"stub:{rank:02x}:00.0", "library_path": __file__, "collectives_posted": None}


class StubEngine:
    name = "stub"
    default_backend = "gloo"
    tensor_device = "cpu"

    def __init__(self, local_rank, args):
        self.local_rank = local_rank
        self.rank = int(os.environ.get("RANK", "0"))
        if os.environ.get("RQ_STUB_DIE_RANK") == str(self.rank):      # a rank that dies before the rendezvous
            raise SystemExit(7)
        if os.environ.get("RQ_STUB_PIDDIR"):                           # lets a test see which processes the launcher started
            with open(os.path.join(os.environ["RQ_STUB_PIDDIR"], f"rank{self.rank}.pid"), "w") as fh:
                fh.write(str(os.getpid()))
        self.last_ms = 0.0
        self.timing = False
        self.t0 = 0.0
        self.shards = []

    def init_process_group(self, dist_module, backend):
        dist_module.init_process_group(backend)

    def make_shard(self, n, offset):
        sh = _StubShard(n, offset)
        self.shards.append(sh)
        return sh

    def synchronize(self):
        pass

    def native_unique_id(self):
        if os.environ.get("RQ_STUB_FAIL_PHASE1") == str(self.rank):
            raise RuntimeError("stub: librccl not found")
        return bytes(range(128))

    def native_exchange(self, world, rank, ident):
        assert ident == bytes(range(128)), "the id every rank uses must be rank 0's, as broadcast"
        if os.environ.get("RQ_STUB_FAIL_PHASE2") == str(self.rank):
            raise RuntimeError("stub: ncclCommInitRank failed")
        return _StubNativeExchange(world)

    def torch_exchange(self, n, n_total, why):
        import bench
        from raptor_amd.distributed import ReturnsExchange
        return bench._TorchExchange(ReturnsExchange(n, n_total, "cpu"), why)

    def local_returns(self, shard):
        return shard.env.finished_returns()

    def set_rollout_timing(self, enable):
        self.timing = bool(enable)

    def last_rollout_ms(self):
        return 1e-3 * 20       # a stand-in kernel time

    def last_rollout_clock_ghz(self):
        return 2.2             # a stand-in core clock

    def timer_start(self):
        self.t0 = time.perf_counter()

    def timer_stop(self):
        return (time.perf_counter() - self.t0) * 1e3

    def describe(self):
        return {"device": "stub (no GPU)", "hip_runtime": None}


def device_count():
    """GPUs the stand-in pretends this node has (bench.spawn_local_ranks refuses more ranks than that)"""
    return int(os.environ.get("RQ_STUB_DEVICES", "8"))


def create_engine(local_rank, args):
    return StubEngine(local_rank, args)
