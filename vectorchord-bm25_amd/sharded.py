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


class RootGather:
    """The gather of the per-rank hit records to rank `dst` (north_star: "top-k gather") with every buffer made ONCE: the
    padded send buffer, the flat receive buffer whose per-rank slices are the gather list, and -- only when the shards differ
    in size -- the compacted output.  Calling it moves the records and allocates nothing (the round-3 version built a list
    of tensors and a torch.cat per step inside the measured loop).

    The tensor a call returns is this object's own receive / output buffer: it is valid until the NEXT call, which overwrites it
    (clone it to keep it)."""

    def __init__(self, n_queries: int, k: int, device, dst: int = 0, group=None):
        import torch
        import torch.distributed as dist

        self.dist, self.group, self.n_queries, self.k = dist, group, n_queries, k
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dst = dst
        self.dst_global = dst if group is None else dist.get_global_rank(group, dst)
        self.per = -(-n_queries // self.world) * k * 3
        lo, hi = shard_bounds(n_queries, self.world, self.rank)
        self.mine = (hi - lo) * k * 3
        self.send = None if self.mine == self.per else torch.zeros(self.per, dtype=torch.int64, device=device)
        self.recv = self.views = self.out = None
        if self.rank == dst:
            self.recv = torch.empty(self.world * self.per, dtype=torch.int64, device=device)
            self.views = list(self.recv.split(self.per))
            if n_queries % self.world:
                self.out = torch.empty(n_queries * k * 3, dtype=torch.int64, device=device)
                self.spans = []
                at = 0
                for r in range(self.world):
                    a, b = shard_bounds(n_queries, self.world, r)
                    n = (b - a) * k * 3
                    self.spans.append((at, r * self.per, n))
                    at += n

    def __call__(self, local_words):
        send = local_words
        if self.send is not None:
            self.send[:self.mine].copy_(local_words[:self.mine])
            send = self.send
        self.dist.gather(send, self.views, dst=self.dst_global, group=self.group)
        if self.rank != self.dst:
            return None
        if self.out is None:
            return self.recv
        for at, src, n in self.spans:
            self.out[at:at + n].copy_(self.recv[src:src + n])
        return self.out


def gather_to_root(local_words, n_queries: int, k: int, dst: int = 0, group=None):
    """One-off form of RootGather (the buffers are made for this one call): the int64 tensor [n_queries * k * 3] in global
    query order on `dst`, None elsewhere.  `dst` is a rank OF `group` (as the shard numbering is)."""
    return RootGather(n_queries, k, local_words.device, dst, group)(local_words)
