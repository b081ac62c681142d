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


def gather_ordered(local, n_total, words_per_elem, group=None, out=None):
    """All-gather per-rank slices (int64 limb tensors, `words_per_elem` limbs per element) into the full
    ordered buffer on every rank.  Even shards (n_total divisible by the world size -- every BASELINE config):
    ONE all_gather_into_tensor straight into the final buffer (rank r's slice lands at r * slice), no staging
    copy.  Uneven shards (not a BASELINE shape): the collective needs equal pieces, so ranks exchange
    max-sized pieces and each is copied to its place.  `out` lets the caller supply the destination (e.g.
    the buffer the commitment hashes)."""
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_total, world)
    if local.numel() != sizes[dist.get_rank(group)] * words_per_elem:
        raise ValueError("local slice has %d words, expected %d" % (local.numel(), sizes[dist.get_rank(group)] * words_per_elem))
    if out is None:
        out = torch.empty(n_total * words_per_elem, dtype=local.dtype, device=local.device)
    if len(set(sizes)) == 1:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    mx = max(sizes) * words_per_elem
    send = local if local.numel() == mx else torch.cat([local, local.new_zeros(mx - local.numel())])
    recv = [torch.empty(mx, dtype=local.dtype, device=local.device) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    off = 0
    for r in range(world):
        cnt = sizes[r] * words_per_elem
        out[off:off + cnt] = recv[r][:cnt]
        off += cnt
    return out


def all_ok(local_ok, device, group=None):
    """AND of the per-shard MAC-verify flags (one word)."""
    t = torch.tensor([1 if local_ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item())
