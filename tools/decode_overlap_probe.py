"""Probe: does the latency-bound decode step overlap with itself?  One model decoding B=32 on one stream vs two model replicas
decoding B=16 each on two streams from two host threads (large-v3-turbo, bf16, random init).  Decides whether a two-branch step graph
(two half-batches sharing one weight copy) is worth building."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlx_swift_audio_amd as m  # noqa: E402
from mlx_swift_audio_amd import synthetic as S, whisper as HW  # noqa: E402

dims = S.DIMS["large-v3-turbo"]
weights = S.synthetic_weights(dims, seed=0, style="survey")
NEW = int(sys.argv[1]) if len(sys.argv) > 1 else 96


def make(B):
    st = torch.cuda.Stream()
    ctx = m.Context(0, stream=st.cuda_stream)
    model = HW.WhisperModel.load(ctx, dims, weights, m.BF16)
    mel = (0.5 * np.random.default_rng(B).standard_normal((B, 2 * dims.n_audio_ctx, dims.n_mels))).astype(np.float32)
    model.encode(mel)
    sp = model.special
    o = HW.DecodingOptions(suppress_ids=S.synthetic_suppress_list(sp), blank_ids=[220], max_new_tokens=NEW)
    return st, ctx, model, o


def run(models):
    ths = [threading.Thread(target=lambda mo=mo, o=o: mo.decode_greedy(o)) for (_, _, mo, o) in models]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


if len(sys.argv) > 2 and sys.argv[2] == "full":
    # two (three) full batches of 32 decoding concurrently on their own streams vs one alone: 1.0x = free overlap, 2.0x (3.0x) = serial
    nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    ms = [make(32) for _ in range(nmax)]
    run(ms[:1])
    t1 = min(run(ms[:1]) for _ in range(2))
    print(f"1 x B=32: {t1*1e3:.1f} ms")
    for n in range(2, nmax + 1):
        run(ms[:n])
        tn = min(run(ms[:n]) for _ in range(2))
        print(f"{n} x B=32 concurrently: {tn*1e3:.1f} ms = {tn/t1:.2f}x of one  ->  {tn/n*1e3:.1f} ms per batch ({n*t1/tn:.2f}x throughput)")
    sys.exit(0)
one = [make(32)]
run(one)
t1 = min(run(one) for _ in range(2))
two = [make(16), make(16)]
run(two)
t2 = min(run(two) for _ in range(2))
half = min(run(two[:1]) for _ in range(2))
steps = NEW + 3
print(f"B=32 one stream : {t1*1e3:8.1f} ms  ({t1/steps*1e6:6.1f} us/step)")
print(f"B=16 alone      : {half*1e3:8.1f} ms  ({half/steps*1e6:6.1f} us/step)")
print(f"2 x B=16 overlap: {t2*1e3:8.1f} ms  ({t2/steps*1e6:6.1f} us/step for 32 clips)  speed-up vs B=32: {t1/t2:.2f}x")
