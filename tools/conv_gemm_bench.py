"""fp32 tap-GEMM timing sweep through mia_op_conv1d_f32 (device buffers): separates fixed launch cost from per-K-tile cost."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlx_swift_audio_amd as M  # noqa: E402

torch.cuda.set_device(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = M.Context(stream=st.cuda_stream)
lib = ctx.lib
lib.mia_op_conv1d_f32.restype = C.c_int
lib.mia_op_conv1d_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 8


def bench(Mr, N, K, taps=1, reps=50):
    Cin = K // taps
    x = torch.randn(Mr + taps, Cin, device="cuda")
    w = torch.randn(N, taps * Cin, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    y = torch.empty(Mr, N, device="cuda")

    def run():
        ctx.check(lib.mia_op_conv1d_f32(ctx.h, x.data_ptr(), Cin, Mr + taps, w.data_ptr(), b.data_ptr(), None, y.data_ptr(), N, Mr, N, Cin, taps, 1, 1, 0, 0))
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"M={Mr:6d} N={N:5d} K={K:5d} taps={taps}: {us:8.1f} us  {2.0 * Mr * N * K / us / 1e6:7.1f} TFLOP/s")
    return us


if __name__ == "__main__":
    for K in (32, 256, 512, 1024, 2048, 4096):
        bench(2100, 256, K)
    for N in (256, 512, 1024, 1536):
        bench(2100, N, 256)
    bench(2100, 256, 768, taps=3)
    bench(36000, 64, 448, taps=7)
    bench(36000, 64, 704, taps=11)
    bench(12000, 128, 896, taps=7)
    bench(48000, 1280, 1280)
