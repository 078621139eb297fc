#!/usr/bin/env python3
"""CosyVoice2 Qwen2LM.inference on one MI355X: Qwen2-0.5B backbone with random-init bf16 weights, an embedding-row
prompt of argv[1] rows (default 300: sos + text + task + prompt speech) and argv[2] generated speech tokens (default 300, EOS
disabled through min_len); argv[3] > 1 adds a run with that many utterances side by side.  Prints one JSON line: prompt-pass time and ms per generated token."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mlx_swift_audio_amd as m
from mlx_swift_audio_amd import lm as HL
from mlx_swift_audio_amd import synthetic as S

n_prompt = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_new = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = S.LM_CONFIGS["qwen2-0.5b"]
ctx = m.Context(0)
w = S.lm_weights(cfg, seed=0, dtype=np.float16)
w.update(S.qwen2lm_extra_weights(cfg, 6561, seed=0))
model = HL.CausalLM.load(ctx, cfg, w, m.BF16)
rng = np.random.default_rng(0)
x = rng.standard_normal((n_prompt, cfg.hidden)).astype(np.float32)
u = rng.random(4 * n_new + 64).astype(np.float32)


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.generate_ras(x, u, n, n, 6561)
    return time.perf_counter() - t0, out


run(8)
d1, _ = run(1)
dt, out = run(n_new)
steps = len(out) - 1
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
batch_res = None
if batch > 1:                                          # utterance-level batching: argv[3] utterances side by side
    model.set_batch(batch)
    xs = [rng.standard_normal((n_prompt, cfg.hidden)).astype(np.float32) for _ in range(batch)]
    ub = rng.random((batch, 4 * n_new + 64)).astype(np.float32)
    model.generate_ras_batch(xs, ub, [8] * batch, [8] * batch, 6561)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = model.generate_ras_batch(xs, ub, [n_new] * batch, [n_new] * batch, 6561)
    db = time.perf_counter() - t0
    ntok = sum(len(o) for o in outs)
    batch_res = {"utterances": batch, "seconds": round(db, 4), "speech_tokens_per_s": round(ntok / db, 1), "audio_seconds_per_second": round(ntok / 25.0 / db, 1)}
    model.set_batch(1)
print(json.dumps({"batch": batch_res, "model": "qwen2-0.5b (Qwen2LM.inference)", "prompt_rows": n_prompt, "generated_tokens": len(out),
                  "prompt_pass_plus_first_step_ms": round(d1 * 1e3, 2), "ms_per_token": round((dt - d1) / max(steps, 1) * 1e3, 3),
                  "seconds": round(dt, 4), "speech_tokens_per_s": round(len(out) / dt, 1),
                  "audio_seconds_per_second": round(len(out) / 25.0 / dt, 2)}))
