"""Multi-GPU sampling: one process per GPU, batch sharded, ONE collective (all-gather of finished images).

The reference has no multi-GPU sampling; samples are independent once their conditioning row
exists, so the batch is split contiguously over ranks, weights and the fp16 DB are replicated, and the only
exchange is an all-gather of the decoded images (RCCL over xGMI on GPUs — backend "nccl" — or gloo in the CPU
tests).  Per-sample RNG streams are a function of (seed, GLOBAL sample index) so results do not depend on the
number of ranks.
"""
import torch
import torch.distributed as dist


def shard_range(n_total: int, world: int, rank: int):
    """Contiguous balanced split: the first n_total % world ranks get one extra sample."""
    q, r = divmod(n_total, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def per_sample_noise(seed: int, global_indices, shape, device="cpu", dtype=torch.float32):
    """x_T for each global sample index from its own generator: invariant to the sharding."""
    out = []
    for gi in global_indices:
        g = torch.Generator(device="cpu").manual_seed((int(seed) * 1_000_003 + int(gi)) % (2 ** 63 - 1))
        out.append(torch.randn(shape, generator=g, dtype=dtype))
    return torch.stack(out).to(device)


def all_gather_images(local: torch.Tensor, n_total: int = None, group=None) -> torch.Tensor:
    """Gather [b_rank, ...] shards into [n_total, ...] on every rank (shards may differ by one sample)."""
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    if n_total is None:
        n_total = local.shape[0] * world
    counts = [shard_range(n_total, world, r)[1] - shard_range(n_total, world, r)[0] for r in range(world)]
    bmax = max(counts)
    if all(c == bmax for c in counts):
        out = torch.empty((world * bmax,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros((bmax,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
