"""Seeded synthetic inputs for the Whisper path: model dimensions, a random-init checkpoint with
the reference's key schema, clips and a stand-in suppress list.  There is no network for real checkpoints or audio, so
the benchmark, the smoke test and the parity tests all draw their data from here (the oracle re-exports these).
This is data generation only -- no reference arithmetic lives here."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

SAMPLE_RATE = 16000


@dataclass
class ModelDimensions:
    """WhisperConfig.swift:9-86"""
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int

    def astuple(self):
        return (self.n_mels, self.n_audio_ctx, self.n_audio_state, self.n_audio_head, self.n_audio_layer,
                self.n_vocab, self.n_text_ctx, self.n_text_state, self.n_text_head, self.n_text_layer)


# public OpenAI dims: not in the reference tree, read from config.json at load time there
DIMS = {
    "tiny.en": ModelDimensions(80, 1500, 384, 6, 4, 51864, 448, 384, 6, 4),
    "large-v3-turbo": ModelDimensions(128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 4),
    "large-v3": ModelDimensions(128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 32),
    # reduced-size layouts for tests (same vocabulary arithmetic as tiny.en / multilingual)
    "micro.en": ModelDimensions(80, 100, 128, 2, 2, 51864, 448, 128, 2, 2),
    "micro": ModelDimensions(128, 100, 128, 2, 2, 51866, 448, 128, 2, 2),
}



def synthetic_suppress_list(special, n: int = 90, seed: int = 11) -> list[int]:
    """The real non-speech list needs the tiktoken vocabulary (WhisperTokenizer.swift:489-532), absent offline:
    a fixed synthetic list of text ids + the specials GreedyDecoder always adds (:190-198)."""
    st = special
    rng = np.random.Generator(np.random.PCG64(seed))
    ids = sorted(int(i) for i in rng.choice(st.eot, size=n, replace=False))
    return ids + [st.transcribe, st.translate, st.sot, st.sot_prev, st.sot_lm, st.no_speech]


def weight_names(d: ModelDimensions) -> dict[str, tuple]:
    """Reference checkpoint schema (Module property paths) -> shapes."""
    D, M = d.n_audio_state, d.n_mels
    out: dict[str, tuple] = {
        "encoder.conv1.weight": (D, 3, M), "encoder.conv1.bias": (D,),
        "encoder.conv2.weight": (D, 3, D), "encoder.conv2.bias": (D,),
        "encoder.ln_post.weight": (D,), "encoder.ln_post.bias": (D,),
        "decoder.token_embedding.weight": (d.n_vocab, D),
        "decoder.positional_embedding": (d.n_text_ctx, D),
        "decoder.ln.weight": (D,), "decoder.ln.bias": (D,),
    }

    def block(p, cross):
        names = {}
        for a in (["attn"] + (["cross_attn"] if cross else [])):
            names[f"{p}.{a}.query.weight"] = (D, D); names[f"{p}.{a}.query.bias"] = (D,)
            names[f"{p}.{a}.key.weight"] = (D, D)
            names[f"{p}.{a}.value.weight"] = (D, D); names[f"{p}.{a}.value.bias"] = (D,)
            names[f"{p}.{a}.out.weight"] = (D, D); names[f"{p}.{a}.out.bias"] = (D,)
            names[f"{p}.{a}_ln.weight"] = (D,); names[f"{p}.{a}_ln.bias"] = (D,)
        names[f"{p}.mlp1.weight"] = (4 * D, D); names[f"{p}.mlp1.bias"] = (4 * D,)
        names[f"{p}.mlp2.weight"] = (D, 4 * D); names[f"{p}.mlp2.bias"] = (D,)
        names[f"{p}.mlp_ln.weight"] = (D,); names[f"{p}.mlp_ln.bias"] = (D,)
        return names

    for l in range(d.n_audio_layer):
        out.update(block(f"encoder.blocks.{l}", False))
    for l in range(d.n_text_layer):
        out.update(block(f"decoder.blocks.{l}", True))
    return out


def _key_seed(name: str, seed: int) -> int:
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return (h + seed * 0x9E3779B1) & 0xFFFFFFFF


def synthetic_weights(d: ModelDimensions, seed: int = 0, style: str = "lecun", round_to: str | None = None) -> dict[str, np.ndarray]:
    """Seeded random-init checkpoint with the reference's key schema.
    style 'survey': N(0, 0.02^2) matrices, LN gamma 1 beta 0.
    style 'lecun' : N(0, 1/fan_in) matrices, small random biases / LN affine -- O(1) activations, harder test.
    round_to: None | 'bf16' | 'f16' rounds every tensor to that storage type (still returned as fp32)."""
    w: dict[str, np.ndarray] = {}
    for name, shape in weight_names(d).items():
        rng = np.random.Generator(np.random.PCG64(_key_seed(name, seed)))
        is_ln = "_ln." in name or ".ln." in name or "ln_post" in name
        if is_ln and name.endswith(".weight"):
            a = np.ones(shape, np.float32) if style == "survey" else (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith(".bias"):
            a = np.zeros(shape, np.float32) if style == "survey" else (0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:
            if style == "survey":
                std = 0.02
            elif name == "decoder.token_embedding.weight":
                std = 1.0 / math.sqrt(shape[-1]) * 4.0     # spread logits: larger argmax margins
            elif name == "decoder.positional_embedding":
                std = 0.02
            else:
                fan_in = int(np.prod(shape[1:]))
                std = 1.0 / math.sqrt(fan_in)
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
        w[name] = round_array(a, round_to)
    return w



def round_array(a: np.ndarray, kind: str | None) -> np.ndarray:
    """Round fp32 values to bf16 / f16 storage precision (returned as fp32)."""
    a = np.ascontiguousarray(a, np.float32)
    if kind is None or kind == "f32":
        return a
    if kind == "bf16":
        u = a.view(np.uint32)
        r = ((u >> 16) & 1) + np.uint32(0x7FFF)
        return (((u + r) >> 16) << 16).astype(np.uint32).view(np.float32)
    if kind == "f16":
        return a.astype(np.float16).astype(np.float32)
    raise ValueError(kind)


def synth_clip(i: int, n_samples: int = 480000) -> np.ndarray:
    """Clip i: 0.1*N(0,1) from PCG64(seed 1000+i) + 0.2*sin(2*pi*220*(1+i%8)*t), clipped to [-1,1]."""
    rng = np.random.Generator(np.random.PCG64(1000 + i))
    t = np.arange(n_samples, dtype=np.float64) / SAMPLE_RATE
    x = 0.1 * rng.standard_normal(n_samples) + 0.2 * np.sin(2 * np.pi * (220.0 * (1 + i % 8)) * t)
    return np.clip(x, -1.0, 1.0).astype(np.float32)
