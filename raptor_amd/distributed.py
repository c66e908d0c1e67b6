"""Multi-GPU sharding of one logical batch of environments.

Environments are independent, so the data path needs no collective: rank r owns the
contiguous global env ids ``shard_range(n_total, world, r)`` and the RNG is keyed by the
GLOBAL id, which makes every per-env result independent of the number of ranks.  The one
exchange is an all-gather of per-env episode returns (RCCL over xGMI when the process group
backend is "nccl"; "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n_total, world_size, rank):
    """Contiguous, balanced partition: the first ``n_total % world_size`` ranks get one extra env."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(int(n_total), int(world_size))
    start = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return start, count


def shard_sizes(n_total, world_size):
    return [shard_range(n_total, world_size, r)[1] for r in range(world_size)]


def all_gather_returns(local_returns, n_total, group=None):
    """Gather per-env values of every rank into one [n_total] tensor ordered by global env id.

    ``local_returns``: 1-D tensor holding this rank's shard (its length must equal
    ``shard_range(n_total, world, rank)[1]``).  Works for uneven shards.
    """
    if not dist.is_available() or not dist.is_initialized():
        if local_returns.numel() != n_total:
            raise ValueError("single process: local shard must be the whole batch")
        return local_returns.clone()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_total, world)
    if local_returns.numel() != sizes[rank]:
        raise ValueError(f"rank {rank}: shard has {local_returns.numel()} envs, expected {sizes[rank]}")
    if len(set(sizes)) == 1:
        out = torch.empty(n_total, dtype=local_returns.dtype, device=local_returns.device)
        dist.all_gather_into_tensor(out, local_returns.contiguous(), group=group)
        return out
    # uneven shards: pad to the largest shard (one fixed-size all-gather), then drop the padding
    width = max(sizes)
    padded = torch.zeros(width, dtype=local_returns.dtype, device=local_returns.device)
    padded[: sizes[rank]] = local_returns
    out = torch.empty(world * width, dtype=local_returns.dtype, device=local_returns.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    out = out.view(world, width)
    return torch.cat([out[r, : sizes[r]] for r in range(world)])


class ReturnsExchange:
    """The path's one exchange, overlapped: a double-buffered all-gather of per-env episode returns that runs
    beside the NEXT rollout instead of between two rollouts.

    ``post(fill)`` hands ``fill`` a local buffer to fill (on a GPU: an asynchronous device-to-device copy
    enqueued on the engine's own HIP stream, e.g. ``env.finished_returns(out=buf, wait=False)``) and starts
    the collective behind it on a side stream; the host does not block.  The engine stream is only held
    back when it is about to overwrite a buffer whose collective (two posts ago) has not finished.
    ``finish()`` blocks until the last collective is done and returns its [n_total] result.
    Equal shards only (``n_total == world * n_local``); ``all_gather_returns`` covers uneven ones.
    """

    def __init__(self, n_local, n_total, device, engine_stream=None, dtype=torch.float32, group=None):
        self.group = group
        self.dist = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size(group) if self.dist else 1
        if n_total != world * n_local:
            raise ValueError("ReturnsExchange needs equal shards")
        device = torch.device(device)
        self.cuda = device.type == "cuda"
        self.bufs = [torch.zeros(n_local, dtype=dtype, device=device) for _ in range(2)]
        self.outs = [torch.zeros(n_total, dtype=dtype, device=device) for _ in range(2)] if self.dist else self.bufs
        self.work = [None, None]
        self.posts = 0
        if self.cuda:
            if engine_stream is None:
                raise ValueError("on a GPU the engine's HIP stream is needed to order the copies")
            self.engine = torch.cuda.ExternalStream(engine_stream, device=device)
            self.side = torch.cuda.Stream(device=device)

    def post(self, fill):
        j = self.posts % 2
        if self.work[j] is not None:           # the collective that last read bufs[j] / wrote outs[j]
            if self.cuda:
                with torch.cuda.stream(self.engine):
                    self.work[j].wait()        # holds the ENGINE stream back, not the host
            else:
                self.work[j].wait()
            self.work[j] = None
        fill(self.bufs[j])
        if self.dist:
            if self.cuda:
                ready = torch.cuda.Event()
                ready.record(self.engine)
                self.side.wait_event(ready)
                with torch.cuda.stream(self.side):
                    self.work[j] = dist.all_gather_into_tensor(self.outs[j], self.bufs[j], group=self.group,
                                                               async_op=True)
            else:
                self.work[j] = dist.all_gather_into_tensor(self.outs[j], self.bufs[j], group=self.group,
                                                           async_op=True)
        self.posts += 1

    def finish(self):
        if self.posts == 0:
            return None
        for j in range(2):
            if self.work[j] is not None:
                if self.cuda:
                    with torch.cuda.stream(self.side):
                        self.work[j].wait()
                else:
                    self.work[j].wait()
                self.work[j] = None
        if self.cuda:
            self.engine.synchronize()
            self.side.synchronize()
        return self.outs[(self.posts - 1) % 2]


class NativeReturnsExchange:
    """The same exchange through the C++ host: ``rq_comm_*`` / ``rq_allgather_returns`` (raptor_quad.h), an RCCL
    all-gather issued by libraptor_quad.so itself on a side stream behind the engine's event, double-buffered.
    This is what a C or C++ host of the library uses; ``bench.py`` uses it too and keeps ``torch.distributed``
    only for the rendezvous (shipping the 128-byte id), the barriers and the max-over-ranks of the timings.

        id_bytes = NativeReturnsExchange.unique_id()     # on rank 0, then broadcast to every rank
        ex = NativeReturnsExchange(device, world, rank, id_bytes)
        ex.post(env) ... ; returns = ex.finish()          # [world * n_envs] float32 numpy, global env order
    """

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.call("rq_comm_unique_id", buf, _lib.COMM_ID_BYTES)
        return bytes(buf.raw)

    def __init__(self, device, world, rank, id_bytes):
        import ctypes as C
        import weakref
        from . import _lib
        if len(id_bytes) != _lib.COMM_ID_BYTES:
            raise ValueError("the communicator id must be 128 bytes")
        h = C.c_void_p()
        _lib.call("rq_comm_create", device._h, int(world), int(rank), C.c_char_p(id_bytes), len(id_bytes), C.byref(h))
        self._h, self._device, self.world, self.rank = h, device, int(world), int(rank)
        self._fin = weakref.finalize(self, _lib.load().rq_comm_destroy, h)
        self.posts = 0

    def info(self):
        """(n_ranks, rank) as the communicator itself reports them (rq_comm_info)."""
        import ctypes as C
        from . import _lib
        n, r = C.c_uint32(), C.c_uint32()
        _lib.call("rq_comm_info", self._h, C.byref(n), C.byref(r))
        return n.value, r.value

    def describe(self):
        """What RCCL and the HIP runtime say about this communicator (rq_comm_describe): ranks, rank, RCCL version, the
        library file the collective's code was mapped from, the device and its PCI bus id -> dict."""
        import ctypes as C
        from . import _lib

        class Description(C.Structure):          # rq_comm_description (include/raptor_quad.h)
            _fields_ = [("struct_bytes", C.c_uint32), ("n_ranks", C.c_uint32), ("rank", C.c_uint32), ("rccl_version", C.c_int32),
                        ("device", C.c_int32), ("pci_bus_id", C.c_char * 32), ("library_path", C.c_char * 256),
                        ("collectives_posted", C.c_uint64)]
        d = Description()
        d.struct_bytes = C.sizeof(Description)
        _lib.call("rq_comm_describe", self._h, C.byref(d))
        v = int(d.rccl_version)
        return {"ranks": int(d.n_ranks), "rank": int(d.rank), "version_code": v,
                "version": None if v < 0 else (f"{v // 10000}.{v // 100 % 100}.{v % 100}" if v >= 10000 else f"{v // 1000}.{v // 100 % 10}.{v % 100}"),
                "device": int(d.device), "pci_bus_id": d.pci_bus_id.decode(), "library_path": d.library_path.decode(),
                "collectives_posted": int(d.collectives_posted)}

    def post(self, env):
        from . import _lib
        _lib.call("rq_allgather_returns", env._require("environment"), self._h)
        self.posts += 1

    def finish(self, to_host=True):
        """Waits for the last posted all-gather.  -> numpy [world * n_envs] (``to_host=False``: (device pointer, count))."""
        import ctypes as C
        import numpy as np
        from . import _lib
        if self.posts == 0:
            return None
        ptr, count = C.c_void_p(), C.c_uint32()
        _lib.call("rq_comm_gathered", self._h, C.byref(ptr), C.byref(count), None)
        if not to_host:
            return ptr.value, count.value
        out = np.empty(count.value, np.float32)
        _lib.call("rq_comm_gathered", self._h, None, None, _lib.fptr(out))
        return out
