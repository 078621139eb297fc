"""Flow-only timing (one CosyVoice2 flow inference, 375 + 150 tokens) -- used under rocprofv3."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlx_swift_audio_amd as M  # noqa: E402
from mlx_swift_audio_amd import flow as HFL, synthetic as S  # noqa: E402

torch.cuda.set_device(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = M.Context(stream=st.cuda_stream)
cfg = S.FLOW_CONFIGS["flow_cosyvoice2"]
fm = HFL.FlowModule.load(ctx, cfg, S.flow_weights(cfg, 0))
rng = np.random.default_rng(0)
n_tok, n_prompt = 375, 150
Tm = 2 * (n_tok + n_prompt)
tok = torch.from_numpy(rng.integers(0, cfg.vocab_size, n_tok).astype(np.int32)).cuda()
ptok = torch.from_numpy(rng.integers(0, cfg.vocab_size, n_prompt).astype(np.int32)).cuda()
pf = torch.randn(2 * n_prompt, 80, device="cuda")
spk = torch.randn(cfg.spk_embed_dim, device="cuda")
z = torch.randn(80, Tm, device="cuda")
mel = torch.empty(80, Tm - 2 * n_prompt, device="cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(reps + 1):
    if i == 1:
        e0.record()
    ctx.check(ctx.lib.mia_flow_inference(fm.h, tok.data_ptr(), n_tok, ptok.data_ptr(), n_prompt, pf.data_ptr(), 2 * n_prompt, spk.data_ptr(), z.data_ptr(), 0,
                                         mel.data_ptr(), 1))
e1.record()
torch.cuda.synchronize()
print(f"flow inference: {e0.elapsed_time(e1) / reps:.2f} ms")

# ---- the same utterance U times through mia_flow_inference_batch (device-resident inputs): mel frames per second of the whole batch
import ctypes as C  # noqa: E402

lib = ctx.lib
lib.mia_flow_inference_batch.restype = C.c_int
lib.mia_flow_inference_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_int]
for U in ((int(sys.argv[2]),) if len(sys.argv) > 2 else (2, 4, 8, 16)):
    arr = lambda v: (C.c_void_p * U)(*([v] * U))
    mels = [torch.empty(80, Tm - 2 * n_prompt, device="cuda") for _ in range(U)]
    outs = (C.c_void_p * U)(*[m_.data_ptr() for m_ in mels])
    n1 = np.full(U, n_tok, np.int32); n2 = np.full(U, n_prompt, np.int32); n3 = np.full(U, 2 * n_prompt, np.int32)
    for i in range(reps + 1):
        if i == 1:
            e0.record()
        ctx.check(lib.mia_flow_inference_batch(fm.h, U, arr(tok.data_ptr()), n1.ctypes.data, arr(ptok.data_ptr()), n2.ctypes.data, arr(pf.data_ptr()),
                                               n3.ctypes.data, arr(spk.data_ptr()), arr(z.data_ptr()), 0, outs, 1))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    same = all(torch.equal(m_, mel) for m_ in mels)
    print(f"flow batch {U:2d}: {ms:8.2f} ms = {ms / U:6.2f} ms per utterance, {U * (Tm - 2 * n_prompt) / ms * 1e3:8.0f} mel frames/s "
          f"({U * (Tm - 2 * n_prompt) / 50.0 / ms * 1e3:6.1f} x real time); every mel equals the single call: {same}")
