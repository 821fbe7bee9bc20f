"""Data parallelism over the (prompt, seed) batch: one process per GPU, full weight replica each, a
contiguous shard of the global batch per rank, NO collective inside the step loop; the only cross-GPU
traffic is one gather of the decoded uint8 images to rank 0 (RCCL on GPUs, gloo in the CPU tests).
The reference has no multi-device code at all; seeds derive from the GLOBAL image
index so results do not depend on the GPU count.
"""
import torch
import torch.distributed as dist


def shard_range(global_batch, rank, world):
    """Contiguous [begin, end) of image indices owned by `rank`; earlier ranks take the remainder."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    per, rem = divmod(global_batch, world)
    begin = rank * per + min(rank, rem)
    return begin, begin + per + (1 if rank < rem else 0)


def latent_for_image(global_index, shape, base_seed=1234):
    """x_T of one image: a function of its global index only (CPU generator: identical on every rank)."""
    g = torch.Generator().manual_seed(base_seed + int(global_index))
    return torch.randn(tuple(shape), generator=g)


def shard_latents(global_batch, rank, world, shape, base_seed=1234):
    b, e = shard_range(global_batch, rank, world)
    return torch.stack([latent_for_image(i, shape, base_seed) for i in range(b, e)]) if e > b \
        else torch.empty((0,) + tuple(shape))


def gather_images(local_u8, global_batch, rank, world, dst=0):
    """Gather per-rank [b_r, H, W, C] uint8 images to `dst` in global-index order.  Ranks may own different
    counts (ragged shards): tensors are padded to the largest shard for the collective and trimmed after."""
    if world == 1:
        return local_u8
    counts = [shard_range(global_batch, r, world) for r in range(world)]
    counts = [e - b for b, e in counts]
    mx = max(counts)
    pad = local_u8
    if local_u8.shape[0] < mx:
        pad = torch.zeros((mx,) + tuple(local_u8.shape[1:]), dtype=local_u8.dtype, device=local_u8.device)
        pad[:local_u8.shape[0]] = local_u8
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r][:counts[r]] for r in range(world)], dim=0)
