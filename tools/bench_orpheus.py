#!/usr/bin/env python3
"""BASELINE.json configs[2]: Orpheus-3B (Llama-3 backbone) autoregression + SNAC decode on one MI355X, random-init bf16 weights,
64-token prompt (argv[3] changes it).  Prints one JSON line: prompt-pass time, ms per generated token, fraction of the HBM roofline
(6.6 GB of weights per token), SNAC samples/s."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mlx_swift_audio_amd as m
from mlx_swift_audio_amd import codec as HC
from mlx_swift_audio_amd import lm as HL
from mlx_swift_audio_amd import synthetic as S

name = sys.argv[1] if len(sys.argv) > 1 else "orpheus-3b"
n_new = int(sys.argv[2]) if len(sys.argv) > 2 else 210
cfg = S.LM_CONFIGS[name]
ctx = m.Context(0)
t0 = time.time()
w = S.lm_weights(cfg, seed=0, dtype=np.float16)
print(f"[orpheus] weights generated in {time.time() - t0:.1f}s", file=sys.stderr, flush=True)
q4 = os.environ.get("MIA_BENCH_Q4") == "1"       # MLX-affine 4-bit step weights (the reference's default Orpheus checkpoint is q4, group 64)
if q4:
    from mlx_swift_audio_amd import checkpoint as CK
    t0 = time.time()
    packed, names = {}, (["model.embed_tokens"] if cfg.tie_embeddings else ["lm_head"])
    for l in range(cfg.n_layers):
        names += [f"model.layers.{l}.self_attn.{n}_proj" for n in "qkvo"] + [f"model.layers.{l}.mlp.{n}_proj" for n in ("gate", "up", "down")]
    for n in names:
        pk, sc, bi = CK.quantize_affine(w[n + ".weight"])
        packed[n + ".weight"], packed[n + ".scales"], packed[n + ".biases"] = pk, sc, bi
        w[n + ".weight"] = CK.dequantize_affine(ctx, pk, sc, bi).astype(np.float16)      # the expanded checkpoint (prompt pass)
    print(f"[orpheus] quantised + expanded in {time.time() - t0:.1f}s", file=sys.stderr, flush=True)
model = HL.CausalLM.load(ctx, cfg, w, m.BF16)
del w
if q4:
    model.attach_q4(packed)
    del packed
scfg = S.SNAC_CONFIGS["snac_24khz"]
snac = HC.SNACDecoder.load(ctx, scfg, S.snac_weights(scfg, 0))
rng = np.random.default_rng(0)
n_prompt = int(sys.argv[3]) if len(sys.argv) > 3 else 64
prompt = rng.integers(0, min(128000, cfg.vocab), n_prompt).tolist()
u = rng.random(n_new).astype(np.float32)
model.generate(prompt, u, max_new_tokens=16, stop_ids=(cfg.vocab - 1,))          # warm-up + graph capture
torch.cuda.synchronize()
t0 = time.perf_counter()
model.generate(prompt, u, max_new_tokens=1, stop_ids=(cfg.vocab - 1,))           # batched prompt pass + the first step
d_prompt = time.perf_counter() - t0
t0 = time.perf_counter()
gen = model.generate(prompt, u, max_new_tokens=n_new, stop_ids=(cfg.vocab - 1,))
dt = time.perf_counter() - t0
steps = len(gen) - 1                     # decode steps after the one the prompt timing already contains
dt_dec = dt - d_prompt
params = sum(int(np.prod(s)) for s in [(cfg.vocab, cfg.hidden)]) + cfg.n_layers * ((cfg.n_heads + 2 * cfg.n_kv_heads) * cfg.head_dim * cfg.hidden +
                                                                              cfg.hidden * cfg.n_heads * cfg.head_dim + 3 * cfg.inter * cfg.hidden)
bytes_per_tok = 2.0 * params if not q4 else params * (0.5 + 4.0 / 64)     # 4 bits + (scale, bias) 2 x 16 bit per 64 weights
n = len(gen) // 7
codes = [rng.integers(0, 4096, n).tolist(), rng.integers(0, 4096, 2 * n).tolist(), rng.integers(0, 4096, 4 * n).tolist()]
noise = rng.standard_normal(snac.noise_len(4 * n)).astype(np.float32)
snac.decode(codes, noise)
t1 = time.perf_counter()
pcm = snac.decode(codes, noise)
ds = time.perf_counter() - t1
# sentence-level batching: argv[4] sequences side by side (same prompt length, own uniforms); weights are read once per step for all
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 0
batch_res = None
if batch > 1:
    model.set_batch(batch)
    prompts = [rng.integers(0, min(128000, cfg.vocab), n_prompt).tolist() for _ in range(batch)]
    ub = rng.random((batch, n_new)).astype(np.float32)
    model.generate_batch(prompts, ub, max_new_tokens=16, stop_ids=(cfg.vocab - 1,))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = model.generate_batch(prompts, ub, max_new_tokens=n_new, stop_ids=(cfg.vocab - 1,))
    db = time.perf_counter() - t0
    ntok = sum(len(o) for o in outs)
    batch_res = {"sequences": batch, "seconds": round(db, 4), "tokens_per_s": round(ntok / db, 1), "ms_per_step": round(db / n_new * 1e3, 3),
                 "audio_seconds_per_second_lm_only": round((ntok / 7 * 2048 / 24000.0) / db, 2)}
    model.set_batch(1)
print(json.dumps({"batch": batch_res, "model": name, "weights": "mlx-affine q4 g64 (packed step)" if q4 else "bf16", "prompt_tokens": n_prompt, "generated_tokens": len(gen), "seconds": round(dt, 4),
                  "prompt_pass_plus_first_step_ms": round(d_prompt * 1e3, 2),
                  "tokens_per_s": round(steps / dt_dec, 1), "ms_per_token": round(dt_dec / steps * 1e3, 3),
                  "weight_GB_per_token": round(bytes_per_tok / 1e9, 3), "hbm_GBs": round(bytes_per_tok * steps / dt_dec / 1e9, 1),
                  "hbm_frac_of_8TBs": round(bytes_per_tok * steps / dt_dec / 8e12, 4),
                  "snac_samples": int(pcm.size), "snac_host_inclusive_ms": round(ds * 1e3, 2),
                  "audio_seconds_per_second_lm_only": round((len(gen) / 7 * 2048 / 24000.0) / dt, 2)}))
