"""N>1 path on CPU: world-size-2 gloo run of the clip sharding + token all-gather used by bench.py (the GPU box runs the
same code over RCCL).  The path has no data-path collective, so this is all there is to test off-GPU."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_tokens(lo, hi, L):
    """Deterministic stand-in for decoder output of global clips [lo, hi)."""
    idx = torch.arange(lo, hi, dtype=torch.int32)
    counts = (idx % 7) + 1
    toks = (idx[:, None] * 1000 + torch.arange(L, dtype=torch.int32)[None, :])
    toks = torch.where(torch.arange(L)[None, :] < counts[:, None], toks, torch.zeros_like(toks))
    return toks.to(torch.int32), counts.to(torch.int32)


def _worker(rank, world, port, n_clips, L, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mlx_swift_audio_amd import parallel as P
    lo, hi = P.shard_range(n_clips, rank, world)
    toks, counts = _fake_tokens(lo, hi, L)
    all_t, all_c = P.gather_tokens(toks, counts, world)
    exp_t, exp_c = _fake_tokens(0, n_clips, L)
    ok = bool(torch.equal(all_t, exp_t) and torch.equal(all_c, exp_c))
    q.put((rank, ok, tuple(all_t.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [64, 5])
def test_shard_and_gather_gloo_world2(n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, 16, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok and shape == (n_clips, 16), (rank, ok, shape)


def test_shard_range_properties():
    sys.path.insert(0, ROOT)
    from mlx_swift_audio_amd import parallel as P
    for n in (0, 1, 31, 32, 256, 257):
        for w in (1, 2, 4, 8):
            spans = [P.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        P.shard_range(4, 2, 2)
