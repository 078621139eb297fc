"""Data-parallel sharding of clips across GPUs of one node.

The path shards by clip: no collective on the data path, full weight replica per rank, ONE exchange per pass -- an
all-gather of the int32 token ids (RCCL over xGMI on the GPU box: torch.distributed backend "nccl"; "gloo" in the CPU
tests).  Nothing here touches the reference's arithmetic.
"""
from __future__ import annotations


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of n_items for `rank`; the first n_items % world ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_tokens(local_tokens, local_counts, world: int, max_shard: int | None = None):
    """All-gather per-clip token rows.  local_tokens [b_local, L] int32, local_counts [b_local] int32 (torch tensors on
    the backend's device).  Shards may be ragged (b_local differs by at most one): rows are padded to max_shard.
    Returns (tokens [sum b, L], counts [sum b]) in global clip order on every rank."""
    import torch
    import torch.distributed as dist

    b_local, L = local_tokens.shape
    if world == 1:
        return local_tokens, local_counts
    sizes = [torch.zeros(1, dtype=torch.int64, device=local_tokens.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([b_local], dtype=torch.int64, device=local_tokens.device))
    sizes = [int(s.item()) for s in sizes]
    cap = max_shard or max(sizes)
    pad_t = torch.zeros((cap, L), dtype=local_tokens.dtype, device=local_tokens.device)
    pad_c = torch.zeros((cap,), dtype=local_counts.dtype, device=local_counts.device)
    pad_t[:b_local] = local_tokens
    pad_c[:b_local] = local_counts
    all_t = [torch.zeros_like(pad_t) for _ in range(world)]
    all_c = [torch.zeros_like(pad_c) for _ in range(world)]
    dist.all_gather(all_t, pad_t)
    dist.all_gather(all_c, pad_c)
    return (torch.cat([t[:n] for t, n in zip(all_t, sizes)]), torch.cat([c[:n] for c, n in zip(all_c, sizes)]))
