#!/usr/bin/env python3
"""CosyVoice2 zero-shot chain at production sizes on one MI355X (random-init weights of the real architectures, synthetic clip):
prepare_conditionals (6 s reference: resample + 128-mel + S3 tokenizer + 80-mel + CAM++ speaker embedding) then synthesize (Qwen2-0.5B RAS loop -> flow
(conformer + 10-step CFM) -> HiFT).  Host-inclusive wall clock per stage (numpy in / numpy out through the ctypes mirror), one
JSON line.  argv[1]: text tokens (default 15 -> up to 300 speech tokens = 12 s)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mlx_swift_audio_amd as m
from mlx_swift_audio_amd import cosyvoice2 as CV, flow as HF, hift as HH, lm as HL, s3tok as HS, speaker as SP
from mlx_swift_audio_amd import synthetic as S

n_text = int(sys.argv[1]) if len(sys.argv) > 1 else 15
ctx = m.Context(0)
lcfg = S.LM_CONFIGS["qwen2-0.5b"]
lw = S.lm_weights(lcfg, seed=0, dtype=np.float16)
lw.update(S.qwen2lm_extra_weights(lcfg, 6561, seed=0))
llm = HL.Qwen2LM(HL.CausalLM.load(ctx, lcfg, lw, m.BF16), lw, speech_token_size=6561)
fcfg = S.FLOW_CONFIGS["flow_cosyvoice2"]
flow = HF.FlowModule.load(ctx, fcfg, S.flow_weights(fcfg, 0))
hcfg = S.HIFT_CONFIGS["hift_cosyvoice2"]
hift = HH.HiFTGenerator.load(ctx, hcfg, S.hift_weights(hcfg, 0))
scfg = S.S3_CONFIGS["s3_v2"]
s3 = HS.S3Tokenizer.load(ctx, scfg, S.s3_weights(scfg, 0))
spk_enc = SP.CAMPlusSpeakerEncoder.load(ctx, S.campplus_weights(0))
model = CV.CosyVoice2Model(ctx, llm, flow, hift, s3, spk_enc)
rng = np.random.default_rng(0)
t = np.arange(6 * 24000, dtype=np.float32) / 24000.0
ref = (0.3 * np.sin(2 * np.pi * 180.0 * t) + 0.05 * rng.standard_normal(t.size)).astype(np.float32)
text = rng.integers(0, 150000, n_text).tolist()
u = rng.random(8000).astype(np.float32)


def timed(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


out = {}
for rep in range(2):                                   # first repetition warms graphs / allocations
    cond, out["prepare_conditionals_ms"] = timed(lambda: model.prepare_conditionals(ref, prompt_text=[1, 2, 3, 4, 5]))
    toks, out["lm_ms"] = timed(lambda: model.generate_tokens(text, cond.prompt_text, cond.prompt_speech_token, u))
    T = 2 * (len(toks) + len(cond.prompt_speech_token))
    z = rng.standard_normal((80, T)).astype(np.float32)
    mel, out["flow_ms"] = timed(lambda: model.tokens_to_mel(np.asarray(toks, np.int32), cond.prompt_speech_token, cond.prompt_mel, cond.speaker_embedding, z))
    noise = rng.standard_normal((mel.shape[1] * hift.up, 9)).astype(np.float32)
    audio, out["hift_ms"] = timed(lambda: model.mel_to_audio(mel, noise))
secs = audio.size / 24000.0
synth = out["lm_ms"] + out["flow_ms"] + out["hift_ms"]
out = {k: round(v, 2) for k, v in out.items()}
out.update({"prompt_speech_tokens": int(len(cond.prompt_speech_token)), "speech_tokens": len(toks), "audio_seconds": round(secs, 2),
            "synthesize_ms": round(synth, 1), "realtime_factor": round(secs / (synth / 1e3), 1)})
print(json.dumps(out))
