#!/usr/bin/env python3
"""What does the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reach on the encoder's shapes?  A yardstick for gemm.hip only:
nothing in the product path calls it.  Prints TFLOP/s per shape (bf16, fp32 accumulate, no epilogue)."""
import time
import torch

dev = torch.device("cuda:0")
shapes = [(48000, 1280, 1280), (48000, 3840, 1280), (48000, 5120, 1280), (48000, 1280, 5120), (48000, 1280, 3840)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"M={M} N={N} K={K}: {dt * 1e3:.3f} ms  {2.0 * M * N * K / dt / 1e12:.0f} TFLOP/s", flush=True)
