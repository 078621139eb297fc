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


def gather_tokens(local_tokens, local_counts, world: int, max_shard: int | None = None, group=None):
    """All-gather per-clip token rows.  local_tokens [b_local, L] int32, local_counts [b_local] int32 (torch tensors on
    the backend's device).  Shards may be ragged (b_local differs by at most one): rows are padded to max_shard.
    Returns (tokens [sum b, L], counts [sum b]) in global clip order on every rank."""
    import torch
    import torch.distributed as dist

    b_local, L = local_tokens.shape
    if world == 1:
        return local_tokens, local_counts
    sizes = [torch.zeros(1, dtype=torch.int64, device=local_tokens.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([b_local], dtype=torch.int64, device=local_tokens.device), group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max_shard or max(sizes)
    pad_t = torch.zeros((cap, L), dtype=local_tokens.dtype, device=local_tokens.device)
    pad_c = torch.zeros((cap,), dtype=local_counts.dtype, device=local_counts.device)
    pad_t[:b_local] = local_tokens
    pad_c[:b_local] = local_counts
    all_t = [torch.zeros_like(pad_t) for _ in range(world)]
    all_c = [torch.zeros_like(pad_c) for _ in range(world)]
    dist.all_gather(all_t, pad_t, group=group)
    dist.all_gather(all_c, pad_c, group=group)
    return (torch.cat([t[:n] for t, n in zip(all_t, sizes)]), torch.cat([c[:n] for c, n in zip(all_c, sizes)]))


# ---- the same exchange through the C ABI (include/mia.h "data-parallel exchange"): RCCL on the context's own stream ----------------
def dp_available(ctx) -> int:
    """1 when RCCL can be bound in this process (mia_dp_available).  mia_dp_init is a collective: vote on this across ranks first."""
    import ctypes as C
    ctx.lib.mia_dp_available.restype = C.c_int
    ctx.lib.mia_dp_available.argtypes = []
    return int(ctx.lib.mia_dp_available())


def dp_unique_id(ctx) -> bytes:
    """Rank 0: the 128-byte RCCL id to hand to every other rank (mia_dp_unique_id)."""
    import ctypes as C
    buf = (C.c_char * 128)()
    ctx.lib.mia_dp_unique_id.argtypes = [C.c_void_p, C.c_void_p]
    ctx.check(ctx.lib.mia_dp_unique_id(ctx.h, buf))
    return bytes(buf)


def dp_init(ctx, rank: int, world: int, unique_id: bytes) -> None:
    import ctypes as C
    ctx.lib.mia_dp_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
    ctx.check(ctx.lib.mia_dp_init(ctx.h, rank, world, unique_id))


def dp_shutdown(ctx) -> None:
    import ctypes as C
    ctx.lib.mia_dp_shutdown.argtypes = [C.c_void_p]
    ctx.check(ctx.lib.mia_dp_shutdown(ctx.h))


def dp_gather_tokens(ctx, local_tokens_ptr: int, local_counts_ptr: int, b_local: int, L: int, n_items: int, all_tokens_ptr: int, all_counts_ptr: int) -> None:
    """mia_dp_gather_tokens on raw device pointers (e.g. torch tensor.data_ptr()); enqueues on the context's stream."""
    import ctypes as C
    ctx.lib.mia_dp_gather_tokens.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    ctx.check(ctx.lib.mia_dp_gather_tokens(ctx.h, local_tokens_ptr, local_counts_ptr, b_local, L, n_items, all_tokens_ptr, all_counts_ptr))
