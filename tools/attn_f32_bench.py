"""fp32 attention timing sweep through mia_op_attention_f32: fixed cost vs per-key-tile cost; checks against torch."""
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
lib.mia_op_attention_f32.restype = C.c_int
lib.mia_op_attention_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float]
for (B, T, H) in ((2, 128, 8), (2, 256, 8), (2, 525, 8), (2, 1050, 8), (2, 2100, 8), (8, 1050, 8)):
    D = H * 64
    qkv = torch.randn(B * T, 3 * D, device="cuda")
    out = torch.empty(B * T, D, device="cuda")

    def run():
        ctx.check(lib.mia_op_attention_f32(ctx.h, qkv.data_ptr(), 3 * D, qkv.data_ptr() + 4 * D, 3 * D, qkv.data_ptr() + 8 * D, 3 * D, out.data_ptr(), D, B, T, H, 0.125))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    q, k, v = [qkv[:, i * D:(i + 1) * D].reshape(B, T, H, 64).permute(0, 2, 1, 3) for i in range(3)]
    ref = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v
    err = (out.reshape(B, T, H, 64).permute(0, 2, 1, 3) - ref).abs().max().item()
    print(f"B={B} T={T:5d} H={H}: {us:8.1f} us  {4.0 * B * H * T * T * 64 / us / 1e6:6.1f} TFLOP/s  workgroups={((T + 31) // 32) * H * B:5d}  maxerr={err:.2e}")
