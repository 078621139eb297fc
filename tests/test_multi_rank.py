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


def test_abi_shard_and_unpack_match_python_rule():
    """The C-ABI side of the exchange (mia_dp_shard_range / mia_dp_shard_cap / mia_dp_unpack_host: host arithmetic, no GPU): same
    shard rule as parallel.shard_range, and unpadding of a simulated all-gather image [world][cap][L] restores global clip order
    for even, ragged, tiny and empty shard sets."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, ROOT)
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import parallel as P
    lib = m._lib.load()
    lib.mia_dp_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mia_dp_unpack_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L = 7
    for n in (0, 1, 5, 32, 256, 257):
        for w in (1, 2, 3, 8):
            cap = lib.mia_dp_shard_cap(n, w)
            assert cap == -(-n // w)
            gathered = np.full((w, max(cap, 1), L), -1, np.int32)      # what ncclAllGather leaves: every rank's padded shard
            for r in range(w):
                lo, hi = C.c_int(), C.c_int()
                assert lib.mia_dp_shard_range(n, r, w, C.byref(lo), C.byref(hi)) == 0
                assert (lo.value, hi.value) == P.shard_range(n, r, w)
                toks, _ = _fake_tokens(lo.value, hi.value, L)
                gathered[r, :hi.value - lo.value] = toks.numpy()
            dense = np.zeros((max(n, 1), L), np.int32)
            g = np.ascontiguousarray(gathered[:, :cap]) if cap else gathered
            assert lib.mia_dp_unpack_host(g.ctypes.data, n, w, L, dense.ctypes.data) == 0
            np.testing.assert_array_equal(dense[:n], _fake_tokens(0, n, L)[0].numpy())
    assert lib.mia_dp_shard_range(4, 2, 2, None, None) != 0          # rank out of range / null outputs are rejected
    assert lib.mia_dp_shard_cap(4, 0) < 0


def test_bench_self_launch_dry_run_world2():
    """`python bench.py --gpus 2 --dry-run` without a launcher: bench.py becomes the launcher (child processes, sysfs device count, no
    HIP), both ranks run 3 replica threads whose 'decodes' finish in jittered order, the rank's ONE exchange thread issues the gathers
    in pass order over gloo, every rank verifies every gathered row, rank 0 prints the contract line with n_gpus 2; every rank must
    exit 0 (the launcher returns the worst exit code)."""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "7", "--warmup", "1", "--batch", "4",
                        "--replicas", "3"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["exchange_order_ok"] is True and d["config"]["passes_exchanged"] == 10 and d["config"]["replicas"] == 3
    # a failing rank must surface: WORLD_SIZE that contradicts --gpus makes every child exit non-zero
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120,
                         env=dict(env, WORLD_SIZE="3", RANK="0"))
    assert bad.returncode != 0


def test_exchanger_issues_in_pass_order():
    """bench.py's Exchanger: submissions arrive out of order from several threads, issue order is 0, 1, 2, ...; an error inside one issue
    does not strand the submitters and is re-raised by close()."""
    import importlib.util
    import random
    import threading
    import time
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = []
    ex = bench.Exchanger(lambda p: seen.append(p))
    ids = list(range(40))
    random.Random(3).shuffle(ids)

    def feed(chunk):
        for i in chunk:
            time.sleep(0.001 * (i % 3))
            ex.submit(i, i)
    ths = [threading.Thread(target=feed, args=(ids[k::4],)) for k in range(4)]
    for t in ths:
        t.start()
    ex.wait_issued(39)
    for t in ths:
        t.join()
    ex.close(40)
    assert seen == list(range(40)) and ex.order == list(range(40))

    def boom(p):
        if p == 2:
            raise RuntimeError("gather failed")
    ex2 = bench.Exchanger(boom)
    for i in range(4):
        ex2.submit(i, i)
    ex2.wait_issued(3)
    with pytest.raises(RuntimeError):
        ex2.close(4)


def test_visible_gpu_count_needs_no_runtime():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = bench.visible_gpu_count()
    assert isinstance(n, int) and n >= 0
    os.environ["HIP_VISIBLE_DEVICES"] = ""
    try:
        assert bench.visible_gpu_count() == 0
    finally:
        del os.environ["HIP_VISIBLE_DEVICES"]


@pytest.mark.gpu
def test_abi_rccl_gather_two_row_lengths(ctx):
    """the unpadding plan is cached under (n_items, world) while the staging layout depends on L -- a second call with another L
    on the same context must still compact correctly (the plan now sits at a fixed offset of the staging buffer)."""
    from mlx_swift_audio_amd import parallel as P
    P.dp_init(ctx, 0, 1, P.dp_unique_id(ctx))
    assert P.dp_available(ctx) == 1
    for L in (448, 16, 448, 7):                        # shrink (no regrow), grow back, shrink again
        toks, counts = _fake_tokens(0, 5, L)
        t_d, c_d = toks.cuda(), counts.cuda()
        out_t, out_c = torch.zeros_like(t_d), torch.zeros_like(c_d)
        P.dp_gather_tokens(ctx, t_d.data_ptr(), c_d.data_ptr(), 5, L, 5, out_t.data_ptr(), out_c.data_ptr())
        ctx.synchronize()
        assert torch.equal(out_t.cpu(), toks) and torch.equal(out_c.cpu(), counts), L
    P.dp_shutdown(ctx)


@pytest.mark.gpu
def test_abi_rccl_gather_world_1(ctx):
    """mia_dp_* on the real RCCL (one rank: communicator set-up, the all-gather on the context's stream, unpadding, shutdown)."""
    import numpy as np
    from mlx_swift_audio_amd import parallel as P
    uid = P.dp_unique_id(ctx)
    assert len(uid) == 128
    P.dp_init(ctx, 0, 1, uid)
    toks, counts = _fake_tokens(0, 5, 16)
    t_d, c_d = toks.cuda(), counts.cuda()
    out_t, out_c = torch.zeros_like(t_d), torch.zeros_like(c_d)
    for _ in range(3):                                                # the plan is cached after the first call
        P.dp_gather_tokens(ctx, t_d.data_ptr(), c_d.data_ptr(), 5, 16, 5, out_t.data_ptr(), out_c.data_ptr())
    ctx.synchronize()
    assert torch.equal(out_t.cpu(), toks) and torch.equal(out_c.cpu(), counts)
    import mlx_swift_audio_amd as m
    with pytest.raises(m.MiaError):
        P.dp_gather_tokens(ctx, t_d.data_ptr(), c_d.data_ptr(), 4, 16, 5, out_t.data_ptr(), out_c.data_ptr())   # wrong shard size
    P.dp_shutdown(ctx)
    with pytest.raises(m.MiaError):
        P.dp_gather_tokens(ctx, t_d.data_ptr(), c_d.data_ptr(), 5, 16, 5, out_t.data_ptr(), out_c.data_ptr())   # no communicator
