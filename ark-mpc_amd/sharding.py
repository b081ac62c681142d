"""Multi-GPU sharding of gate batches: one process per GPU, contiguous index ranges,
NO collective in the arithmetic.  The only exchanges are (i) an all-gather of opened-value / MAC-check
buffers so the (inherently sequential) SHA3 commitment and the caller see one ordered buffer, and (ii) an
all-reduce(AND) of the one-word MAC-verify flag.  Backend "nccl" is RCCL over xGMI on ROCm; the same code
runs on "gloo" for the CPU tests."""
import torch
import torch.distributed as dist


def shard_range(n, world, rank):
    """Contiguous range [lo, hi) of gate indices owned by `rank`: [rank*n/world, (rank+1)*n/world)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def shard_sizes(n, world):
    return [shard_range(n, world, r)[1] - shard_range(n, world, r)[0] for r in range(world)]


def gather_ordered(local, n_total, words_per_elem, group=None):
    """All-gather per-rank slices (int64 limb tensors, `words_per_elem` limbs per element) into the full
    ordered buffer on every rank.  Uneven shards are padded to the largest shard for the collective."""
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_total, world)
    mx = max(sizes) * words_per_elem
    pad = torch.zeros(mx, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local
    out = torch.empty(world * mx, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = [out[r * mx: r * mx + sizes[r] * words_per_elem] for r in range(world)]
    return torch.cat(parts)


def all_ok(local_ok, device, group=None):
    """AND of the per-shard MAC-verify flags (one word)."""
    t = torch.tensor([1 if local_ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item())
