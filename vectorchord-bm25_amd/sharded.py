"""Query-batch data parallelism over the GPUs of one node (one process per GPU).

The index is replicated; independent queries are split across ranks; the only exchange is the
gather of the per-rank top-k records (24 bytes per hit), done with torch.distributed
(backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).  No collective touches postings.
"""
import numpy as np


def shard_bounds(n_queries: int, world: int, rank: int):
    """Contiguous, balanced split of query indices: [lo, hi) of `rank`."""
    base, rem = divmod(n_queries, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_queries(term_ids, q_off, world: int, rank: int):
    """CSR slice (term_ids, q_off) of the queries owned by `rank`."""
    q_off = np.asarray(q_off)
    lo, hi = shard_bounds(len(q_off) - 1, world, rank)
    return (np.ascontiguousarray(term_ids[q_off[lo]:q_off[hi]], dtype=np.uint32),
            (q_off[lo:hi + 1] - q_off[lo]).astype(np.uint32))


def gather_hits(local_words, n_queries: int, k: int, group=None):
    """All-gather the per-rank hit records.

    local_words: torch int64 tensor viewing this rank's hits (3 words per 24-byte record,
    `hi - lo` queries x k records), on the device of the backend.  Shards may differ by one
    query, so every rank contributes a buffer padded to the largest shard.  Returns an int64
    tensor [n_queries * k * 3] with the records in global query order."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per = -(-n_queries // world) * k * 3
    send = local_words
    if send.numel() != per:
        send = torch.zeros(per, dtype=torch.int64, device=local_words.device)
        send[:local_words.numel()] = local_words
    out = torch.empty(world * per, dtype=torch.int64, device=local_words.device)
    dist.all_gather_into_tensor(out, send, group=group)
    if n_queries % world == 0:
        return out
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_queries, world, r)
        parts.append(out[r * per:r * per + (hi - lo) * k * 3])
    return torch.cat(parts)


def gather_to_root(local_words, n_queries: int, k: int, dst: int = 0, group=None):
    """Gather the per-rank hit records to rank `dst` only (north_star: "top-k gather"): the all-gather of
    gather_hits moves world x the bytes anybody needs.  Same padding rule; returns the int64 tensor
    [n_queries * k * 3] in global query order on `dst`, None elsewhere.  `dst` is a rank OF `group` (as the
    shard numbering is); torch.distributed.gather wants the global rank, so it is translated here."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = -(-n_queries // world) * k * 3
    send = local_words
    if send.numel() != per:
        send = torch.zeros(per, dtype=torch.int64, device=local_words.device)
        send[:local_words.numel()] = local_words
    bufs = [torch.empty(per, dtype=torch.int64, device=local_words.device) for _ in range(world)] if rank == dst else None
    dst_global = dst if group is None else dist.get_global_rank(group, dst)
    dist.gather(send, bufs, dst=dst_global, group=group)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_queries, world, r)
        parts.append(bufs[r][:(hi - lo) * k * 3])
    return torch.cat(parts)
