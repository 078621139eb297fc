#!/usr/bin/env python3
"""CosyVoice2 flow inferences of several utterances at once: argv[1] worker threads, each with its own mia context (HIP stream) and
FlowModule handle, running `mia_flow_inference` (375 + 150 tokens, 10 Euler steps) back to back.  One inference leaves most of the
chip idle (M = 2 100-row fp32 GEMMs, 130-530 workgroups), so independent utterances overlap.  Prints aggregate mel frames/s."""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mlx_swift_audio_amd as M
from mlx_swift_audio_amd import flow as HFL, synthetic as S

n_workers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_device(0)
cfg = S.FLOW_CONFIGS["flow_cosyvoice2"]
w = S.flow_weights(cfg, 0)
n_tok, n_prompt = 375, 150
Tm = 2 * (n_tok + n_prompt)
rng = np.random.default_rng(0)


class Worker:
    def __init__(self):
        self.stream = torch.cuda.Stream()
        self.ctx = M.Context(stream=self.stream.cuda_stream)
        self.fm = HFL.FlowModule.load(self.ctx, cfg, w)
        with torch.cuda.stream(self.stream):
            self.tok = torch.from_numpy(rng.integers(0, cfg.vocab_size, n_tok).astype(np.int32)).cuda()
            self.ptok = torch.from_numpy(rng.integers(0, cfg.vocab_size, n_prompt).astype(np.int32)).cuda()
            self.pf = torch.randn(2 * n_prompt, 80, device="cuda")
            self.spk = torch.randn(cfg.spk_embed_dim, device="cuda")
            self.z = torch.randn(80, Tm, device="cuda")
            self.mel = torch.empty(80, Tm - 2 * n_prompt, device="cuda")
        self.stream.synchronize()

    def run(self, n):
        for _ in range(n):
            self.ctx.check(self.ctx.lib.mia_flow_inference(self.fm.h, self.tok.data_ptr(), n_tok, self.ptok.data_ptr(), n_prompt, self.pf.data_ptr(), 2 * n_prompt,
                                                           self.spk.data_ptr(), self.z.data_ptr(), 0, self.mel.data_ptr(), 1))
        self.stream.synchronize()


def timed(workers, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=wk.run, args=(n,)) for wk in workers]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


workers = [Worker() for _ in range(n_workers)]
timed(workers, 1)
d1 = timed(workers[:1], reps)
dn = timed(workers, reps)
frames = Tm - 2 * n_prompt
print(json.dumps({"workers": n_workers, "one_worker_ms_per_inference": round(d1 / reps * 1e3, 1),
                  "all_workers_ms_per_inference": round(dn / (reps * n_workers) * 1e3, 1),
                  "mel_frames_per_s_one": round(frames * reps / d1), "mel_frames_per_s_all": round(frames * reps * n_workers / dn),
                  "realtime_factor_all": round(frames * reps * n_workers / dn / 50.0, 1)}))
