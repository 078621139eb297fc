#!/usr/bin/env python3
"""A/B timing of GEMM variants in ONE process (interleaved rounds, random data), encoder shapes of large-v3-turbo b=32."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mlx_swift_audio_amd as m
from mlx_swift_audio_amd import ops

stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = m.Context(0, stream=stream.cuda_stream)
ops._declare(ctx.lib)
shapes = [(48000, 3840, 1280), (48000, 1280, 1280), (48000, 5120, 1280), (48000, 1280, 5120)]
variants = [int(v) for v in (sys.argv[1:] or ["0", "1"])]
for (M, N, K) in shapes:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {v: [] for v in variants}
    for rnd in range(6):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ctx.check(ctx.lib.mia_op_linear(ctx.h, x.data_ptr(), K, w.data_ptr(), None, None, 0, y.data_ptr(), N, M, N, K, 0,
                                                m.BF16, 0, v, m._lib.MEM_DEVICE))
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[v].append(e0.elapsed_time(e1) / 5)
    err = 0.0     # full-output check of the LAST variant run, in row chunks (fp32 reference of the same bf16 operands)
    for r0 in range(0, M, 8192):
        ref = (x[r0:r0 + 8192].float() @ w.float().t())
        err = max(err, (y[r0:r0 + 8192].float() - ref).abs().max().item())
    for v in variants:
        ms = float(np.median(res[v]))
        print(f"M={M} N={N} K={K} variant={v}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s  (min {min(res[v]):.3f})  maxerr(last)={err:.3g}", flush=True)
