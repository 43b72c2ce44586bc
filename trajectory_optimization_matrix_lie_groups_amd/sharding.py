"""Batch sharding across ranks (one process per GPU) and the final gather.

Trajectories are independent (the reference runs them in separate joblib workers,
visualization/perturb_all_compute.py:240-250), so the only collective of the whole path is one
all_gather of results at the end: torch.distributed, backend "nccl" (= RCCL over xGMI) on GPUs,
"gloo" in the CPU tests."""
import torch
import torch.distributed as dist


def shard_bounds(B, world, rank):
    """Contiguous partition [lo, hi) of B trajectories; the first B % world ranks get one more."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(local: torch.Tensor, B: int, group=None) -> torch.Tensor:
    """all_gather of per-trajectory results along dim 0 with uneven shards (padded to the largest)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(B, world, r)[1] - shard_bounds(B, world, r)[0] for r in range(world)]
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[:n] for o, n in zip(outs, sizes)], dim=0)
