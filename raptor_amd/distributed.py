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
