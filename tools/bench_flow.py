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
