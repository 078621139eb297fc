#!/usr/bin/env python3
"""One LM decode run for profiling (rocprofv3 --kernel-trace --stats -- python3 tools/lm_step_profile.py orpheus-3b q4|bf16 [n_new] [layers]):
random-init bf16 checkpoint + random packed MLX-q4 words (bench.py's lm leg), 64-token prompt, n_new sampled tokens, one sequence."""
import dataclasses
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import mlx_swift_audio_amd as m
from mlx_swift_audio_amd import lm as HL
from mlx_swift_audio_amd import synthetic as S

name, mode = sys.argv[1], sys.argv[2]
n_new = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = S.LM_CONFIGS[name]
if len(sys.argv) > 4:
    cfg = dataclasses.replace(cfg, n_layers=int(sys.argv[4]))
ctx = m.Context(0)
w = S.lm_weights(cfg, seed=0, dtype=np.float16)
model = HL.CausalLM.load(ctx, cfg, w, m.BF16)
rng = np.random.default_rng(0)
if mode == "q4":
    lin = (["model.embed_tokens"] if cfg.tie_embeddings else ["lm_head"])
    for l in range(cfg.n_layers):
        lin += [f"model.layers.{l}.self_attn.{n}_proj" for n in "qkvo"] + [f"model.layers.{l}.mlp.{n}_proj" for n in ("gate", "up", "down")]
    packed = {}
    for n in lin:
        N, K = w[n + ".weight"].shape
        packed[n + ".weight"] = rng.integers(0, 1 << 32, (N, K // 8), dtype=np.uint32)
        packed[n + ".scales"] = np.full((N, K // 64), 0.004, np.float16)
        packed[n + ".biases"] = np.full((N, K // 64), -0.03, np.float16)
    model.attach_q4(packed)
    del packed
del w
stop = (cfg.vocab - 1,)
prompt = rng.integers(0, min(128000, cfg.vocab), 64).tolist()
u = rng.random(n_new).astype(np.float32)
model.generate(prompt, u, max_new_tokens=8, stop_ids=stop)
ctx.synchronize()
t0 = time.perf_counter()
model.generate(prompt, u, max_new_tokens=1, stop_ids=stop)
d0 = time.perf_counter() - t0
t0 = time.perf_counter()
gen = model.generate(prompt, u, max_new_tokens=n_new, stop_ids=stop)
dt = time.perf_counter() - t0
print(f"{name} {mode} layers {cfg.n_layers}: {len(gen)} tokens, {(dt - d0) / max(len(gen) - 1, 1) * 1e3:.4f} ms/token (prompt pass + first step {d0 * 1e3:.2f} ms)")
model.close()
