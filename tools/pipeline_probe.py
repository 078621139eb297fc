"""Probe: R model replicas on R HIP streams, passes dealt round-robin from R host threads -- does the encoder of one batch overlap the
(latency-bound) decoder of another?  large-v3-turbo bf16, 32 clips per pass, full 448-token budget."""
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
B = 32
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
clips = np.stack([S.synth_clip(i) for i in range(B)])
offs = np.arange(B + 1, dtype=np.int64) * clips.shape[1]


def make():
    st = torch.cuda.Stream()
    ctx = m.Context(0, stream=st.cuda_stream)
    model = HW.WhisperModel.load(ctx, dims, weights, m.BF16)
    with torch.cuda.stream(st):
        pcm = torch.from_numpy(clips).cuda()
        tok = torch.zeros((B, 448), dtype=torch.int32, device="cuda")
        n = torch.zeros(B, dtype=torch.int32, device="cuda")
        a = torch.zeros(B, dtype=torch.float32, device="cuda")
        ns = torch.zeros(B, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    o = HW.DecodingOptions(suppress_ids=S.synthetic_suppress_list(model.special), blank_ids=[220])

    def step():
        model.transcribe_windows_device(pcm.data_ptr(), offs, o, tok.data_ptr(), n.data_ptr(), a.data_ptr(), ns.data_ptr())
    return step, ctx


def run(steps, k):
    def worker(fn, cnt):
        for _ in range(cnt):
            fn()
    ths = [threading.Thread(target=worker, args=(fn, k // len(steps) + (1 if i < k % len(steps) else 0))) for i, (fn, _) in enumerate(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for _, ctx in steps:
        ctx.synchronize()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for R in (1, 2, 3):
    reps = [make() for _ in range(R)]
    run(reps, R)                    # warm-up (graph capture)
    t = min(run(reps, K) for _ in range(2))
    print(f"replicas={R}: {K} passes in {t*1e3:8.1f} ms -> {30.0 * B * K / t:8.1f} audio-s/s", flush=True)
    del reps
