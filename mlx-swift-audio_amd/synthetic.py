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
    style 'peaky' : 'lecun' re-balanced so that greedy decoding is NOT degenerate (with tied embeddings a random checkpoint repeats one
                    token: the residual stream is dominated by the embedding of the token just fed, whose own logit |e_t|^2 then wins):
                    the final decoder LayerNorm gain gets a random sign per channel (the tied projection of e_t onto itself becomes a
                    zero-mean sum), embedding rows are unit vectors with log-normal gains (a heavy-tailed logit distribution: top-2
                    margins far above the 16-bit rounding noise, softmax peaked enough that the timestamp heuristic does not mask every
                    step), decoder query / key matrices x2 (peaked attention: the output depends on WHICH frames / earlier tokens are
                    attended), positional rows N(0, 0.3^2), conv1 x6 (audio content outweighs the sinusoidal positions).  The result
                    depends on the clip, the position and the token history; tests assert that on the oracle's run.
    round_to: None | 'bf16' | 'f16' rounds every tensor to that storage type (still returned as fp32)."""
    w: dict[str, np.ndarray] = {}
    peaky = style == "peaky"
    for name, shape in weight_names(d).items():
        rng = np.random.Generator(np.random.PCG64(_key_seed(name, seed)))
        is_ln = "_ln." in name or ".ln." in name or "ln_post" in name
        if is_ln and name.endswith(".weight"):
            a = np.ones(shape, np.float32) if style == "survey" else (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
            if peaky and name == "decoder.ln.weight":
                a = a * np.where(rng.random(shape) < 0.5, -1.0, 1.0).astype(np.float32)
        elif name.endswith(".bias"):
            a = np.zeros(shape, np.float32) if style == "survey" else (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif peaky and name == "decoder.token_embedding.weight":
            a = rng.standard_normal(shape, dtype=np.float32)
            a /= np.linalg.norm(a, axis=1, keepdims=True)
            a *= np.exp(0.5 * rng.standard_normal(shape[0])).astype(np.float32)[:, None]
        else:
            if style == "survey":
                std = 0.02
            elif name == "decoder.token_embedding.weight":
                std = 1.0 / math.sqrt(shape[-1]) * 4.0     # spread logits: larger argmax margins
            elif name == "decoder.positional_embedding":
                std = 0.3 if peaky else 0.02
            else:
                fan_in = int(np.prod(shape[1:]))
                std = 1.0 / math.sqrt(fan_in)
                if peaky and name.startswith("decoder.") and (name.endswith("query.weight") or name.endswith("key.weight")):
                    std *= 2.0
                if peaky and name == "encoder.conv1.weight":
                    std *= 6.0
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


# ---- neural codecs (SNAC 24 kHz, DAC speech 24 kHz) -------------------------------------------------------------
@dataclass
class SNACConfig:
    """SNACConfig.swift:9-94 defaults == mlx-community/snac_24khz."""
    latent_dim: int = 768
    decoder_dim: int = 1024
    decoder_rates: tuple = (8, 8, 4, 2)
    vq_strides: tuple = (4, 2, 1)
    codebook_size: int = 4096
    codebook_dim: int = 8
    noise: bool = True
    depthwise: bool = True
    noise_channels: int = 1      # NoiseBlock.swift:18-24 builds 1 output channel; upstream checkpoints carry dim -> dim


@dataclass
class DACConfig:
    """DACConfig.speechDefault (DACModel.swift:192-202), decoder side; latent = encoder_dim * 2^len(rates) = 1024."""
    latent_dim: int = 1024
    decoder_dim: int = 1536
    decoder_rates: tuple = (8, 5, 4, 2)
    n_codebooks: int = 2
    codebook_size: int = 1024
    codebook_dim: int = 8
    encoder_dim: int = 64                      # DACEncoder(dModel:) ; latent = encoder_dim * 2^len(encoder_rates)
    encoder_rates: tuple = (2, 4, 5, 8)        # hop length 320 (DACModel.swift:192-202)


SNAC_CONFIGS = {"snac_24khz": SNACConfig(), "snac_micro": SNACConfig(64, 128, (4, 2), (2, 1), 64, 8), "snac_micro_cn": SNACConfig(64, 128, (4, 2), (2, 1), 64, 8, noise_channels=-1)}
DAC_CONFIGS = {"dac_speech": DACConfig(), "dac_micro": DACConfig(64, 128, (4, 5), 2, 64, 8, 32, (4,))}


def _wn_pair(rng, shape, norm_axes, g_shape):
    """A (weight_g, weight_v) pair as a checkpoint stores them: v arbitrary, g ~ the norm of a LeCun-scaled weight."""
    fan_in = int(np.prod([shape[a] for a in norm_axes]))
    v = rng.standard_normal(shape, dtype=np.float32) / np.float32(math.sqrt(max(fan_in, 1)))
    g = (np.float32(1.0) + np.float32(0.1) * rng.standard_normal(g_shape, dtype=np.float32)).astype(np.float32)
    return g, v.astype(np.float32)


def snac_weights(cfg: SNACConfig, seed: int = 0) -> dict[str, np.ndarray]:
    """Random-init SNAC decoder + quantizer tensors in the checkpoint key schema (SNACDecoder.swift:101-243,358-368)."""
    w: dict[str, np.ndarray] = {}
    rng = np.random.Generator(np.random.PCG64(seed + 77))
    P = "decoder.model.layers."

    def conv(p, cout, k, cin_g, bias=True):
        g, v = _wn_pair(rng, (cout, k, cin_g), (1, 2), (cout, 1, 1))
        w[p + ".weight_g"], w[p + ".weight_v"] = g, v
        if bias:
            w[p + ".bias"] = (0.05 * rng.standard_normal(cout)).astype(np.float32)

    def alpha(p, c):
        w[p] = (1.0 + 0.2 * rng.standard_normal((1, c, 1))).astype(np.float32).clip(0.3, 2.0)

    for i in range(len(cfg.vq_strides)):
        q = f"quantizer.quantizers.{i}"
        w[q + ".codebook.weight"] = rng.standard_normal((cfg.codebook_size, cfg.codebook_dim), dtype=np.float32)
        g, v = _wn_pair(rng, (cfg.latent_dim, 1, cfg.codebook_dim), (1, 2), (cfg.latent_dim, 1, 1))
        w[q + ".out_proj.weight_g"], w[q + ".out_proj.weight_v"] = g, v
        w[q + ".out_proj.bias"] = (0.05 * rng.standard_normal(cfg.latent_dim)).astype(np.float32)
    conv(P + "0", cfg.latent_dim, 7, 1)
    conv(P + "1", cfg.decoder_dim, 1, cfg.latent_dim)
    cin = cfg.decoder_dim
    for i, s in enumerate(cfg.decoder_rates):
        cout = cfg.decoder_dim >> (i + 1)
        b = f"{P}{2 + i}.block.layers."
        alpha(b + "0.alpha", cin)
        g, v = _wn_pair(rng, (cin, 2 * s, cout), (1, 2), (cin, 1, 1))
        w[b + "1.weight_g"], w[b + "1.weight_v"] = g, v
        w[b + "1.bias"] = (0.05 * rng.standard_normal(cout)).astype(np.float32)
        ru = 2
        if cfg.noise:
            cn = cout if cfg.noise_channels < 0 else cfg.noise_channels
            g, v = _wn_pair(rng, (cn, 1, cout), (1, 2), (cn, 1, 1))
            w[b + "2.linear.weight_g"], w[b + "2.linear.weight_v"] = g, v
            ru = 3
        for r in range(3):
            u = f"{b}{ru + r}.block.layers."
            alpha(u + "0.alpha", cout)
            conv(u[:-1] + ".1", cout, 7, 1)
            alpha(u + "2.alpha", cout)
            conv(u[:-1] + ".3", cout, 1, cout)
        cin = cout
    n = len(cfg.decoder_rates)
    alpha(f"{P}{2 + n}.alpha", cin)
    conv(f"{P}{3 + n}", 1, 7, cin)
    return w


def dac_weights(cfg: DACConfig, seed: int = 0) -> dict[str, np.ndarray]:
    """Random-init DAC decoder + quantizer tensors with the reference's Module key paths (DACModel.swift:120-164, DACLayers.swift)."""
    w: dict[str, np.ndarray] = {}
    rng = np.random.Generator(np.random.PCG64(seed + 99))
    P = "decoder.model.layers."

    def conv(p, cout, k, cin):
        g, v = _wn_pair(rng, (cout, k, cin), (1, 2), (cout, 1, 1))
        w[p + ".weight_g"], w[p + ".weight_v"] = g, v
        w[p + ".bias"] = (0.05 * rng.standard_normal(cout)).astype(np.float32)

    def alpha(p, c):
        w[p] = (1.0 + 0.2 * rng.standard_normal((1, 1, c))).astype(np.float32).clip(0.3, 2.0)

    for i in range(cfg.n_codebooks):
        q = f"quantizer.quantizers.{i}"
        w[q + ".codebook.weight"] = rng.standard_normal((cfg.codebook_size, cfg.codebook_dim), dtype=np.float32)
        conv(q + ".out_proj", cfg.latent_dim, 1, cfg.codebook_dim)
    conv(P + "0", cfg.decoder_dim, 7, cfg.latent_dim)
    cin = cfg.decoder_dim
    for i, s in enumerate(cfg.decoder_rates):
        cout = cfg.decoder_dim >> (i + 1)
        b = f"{P}{1 + i}.block.layers."
        alpha(b + "0.alpha", cin)
        g, v = _wn_pair(rng, (cout, 2 * s, cin), (0, 1), (1, 1, cin))      # normalised per INPUT channel (exceptDim 2)
        w[b + "1.weight_g"], w[b + "1.weight_v"] = g, v
        w[b + "1.bias"] = (0.05 * rng.standard_normal(cout)).astype(np.float32)
        for r in range(3):
            u = f"{b}{2 + r}.block.layers."
            alpha(u + "0.alpha", cout)
            conv(u[:-1] + ".1", cout, 7, cout)
            alpha(u + "2.alpha", cout)
            conv(u[:-1] + ".3", cout, 1, cout)
        cin = cout
    n = len(cfg.decoder_rates)
    alpha(f"{P}{1 + n}.alpha", cin)
    conv(f"{P}{2 + n}", 1, 7, cin)
    # ---- encoder (DACModel.swift:13-86) + the quantizers' in_proj (DACQuantize.swift:37-41); drawn AFTER the decoder tensors so that
    # the decoder-side values of a given seed are the ones earlier fixtures were made with
    E = "encoder.block.layers."
    conv(E + "0", cfg.encoder_dim, 7, 1)
    c = cfg.encoder_dim
    for i, st in enumerate(cfg.encoder_rates):
        b = f"{E}{1 + i}.block.layers."
        for r in range(3):
            u = f"{b}{r}.block.layers."
            alpha(u + "0.alpha", c)
            conv(u[:-1] + ".1", c, 7, c)
            alpha(u + "2.alpha", c)
            conv(u[:-1] + ".3", c, 1, c)
        alpha(b + "3.alpha", c)
        conv(b + "4", 2 * c, 2 * st, c)
        c *= 2
    ne = len(cfg.encoder_rates)
    alpha(f"{E}{1 + ne}.alpha", c)
    conv(f"{E}{2 + ne}", cfg.latent_dim, 3, c)
    for i in range(cfg.n_codebooks):
        conv(f"quantizer.quantizers.{i}.in_proj", cfg.codebook_dim, 1, cfg.latent_dim)
    return w


# ---- autoregressive LMs ------------------------------------------------------------------------------------------
@dataclass
class LMConfig:
    vocab: int
    hidden: int
    inter: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    head_dim: int
    max_ctx: int = 2048
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_llama3: bool = True
    rope_factor: float = 32.0
    rope_low: float = 1.0
    rope_high: float = 4.0
    rope_old_ctx: int = 8192
    qkv_bias: bool = False
    tie_embeddings: bool = True


LM_CONFIGS = {
    # OrpheusConfig (TransformerBlock.swift:16-34) == Llama-3.2-3B with the Orpheus vocabulary
    "orpheus-3b": LMConfig(156940, 3072, 8192, 28, 24, 8, 128, 2048),
    # Qwen2Config defaults (Qwen2LM.swift:15-43): the CosyVoice2-0.5B backbone
    "qwen2-0.5b": LMConfig(151936, 896, 4864, 24, 14, 2, 64, 2048, 1e-6, 1e6, False, 1.0, 1.0, 4.0, 8192, True, True),
    # mid-size shape for the packed-q4 step: K = 1024 / 2048 give several 128-input blocks per wave, 4-wave workgroups and cross-workgroup splits
    "llama-q4mid": LMConfig(3000, 1024, 2048, 2, 8, 2, 128, 256),
    "llama-micro": LMConfig(3000, 256, 512, 2, 4, 2, 64, 256),
    "llama-micro128": LMConfig(3000, 256, 512, 2, 2, 1, 128, 256),
    "qwen-micro": LMConfig(3000, 128, 384, 2, 2, 1, 64, 256, 1e-6, 1e6, False, 1.0, 1.0, 4.0, 8192, True, True),
}


def lm_weights(cfg: LMConfig, seed: int = 0, round_to: str | None = None, dtype=np.float32) -> dict[str, np.ndarray]:
    """Random-init causal-LM checkpoint (HF key schema).  N(0, 1/fan_in) matrices, RMSNorm weights ~ 1, embeddings N(0,1)
    scaled so the tied logits have a usable spread."""
    w: dict[str, np.ndarray] = {}
    D, dh = cfg.hidden, cfg.head_dim

    def mat(name, rows, cols, std=None):
        rng = np.random.Generator(np.random.PCG64(_key_seed(name, seed)))
        a = rng.standard_normal((rows, cols), dtype=np.float32) * np.float32(std if std is not None else 1.0 / math.sqrt(cols))
        w[name] = round_array(a, round_to).astype(dtype)

    def vec(name, n, base, jitter):
        rng = np.random.Generator(np.random.PCG64(_key_seed(name, seed)))
        w[name] = round_array((base + jitter * rng.standard_normal(n)).astype(np.float32), None)

    mat("model.embed_tokens.weight", cfg.vocab, D, 1.0 / math.sqrt(D) * 3.0)
    if not cfg.tie_embeddings:
        mat("lm_head.weight", cfg.vocab, D)
    vec("model.norm.weight", D, 1.0, 0.1)
    for l in range(cfg.n_layers):
        p = f"model.layers.{l}"
        vec(p + ".input_layernorm.weight", D, 1.0, 0.1)
        vec(p + ".post_attention_layernorm.weight", D, 1.0, 0.1)
        mat(p + ".self_attn.q_proj.weight", cfg.n_heads * dh, D)
        mat(p + ".self_attn.k_proj.weight", cfg.n_kv_heads * dh, D)
        mat(p + ".self_attn.v_proj.weight", cfg.n_kv_heads * dh, D)
        mat(p + ".self_attn.o_proj.weight", D, cfg.n_heads * dh)
        if cfg.qkv_bias:
            vec(p + ".self_attn.q_proj.bias", cfg.n_heads * dh, 0.0, 0.1)
            vec(p + ".self_attn.k_proj.bias", cfg.n_kv_heads * dh, 0.0, 0.1)
            vec(p + ".self_attn.v_proj.bias", cfg.n_kv_heads * dh, 0.0, 0.1)
        mat(p + ".mlp.gate_proj.weight", cfg.inter, D)
        mat(p + ".mlp.up_proj.weight", cfg.inter, D)
        mat(p + ".mlp.down_proj.weight", D, cfg.inter)
    return w


# ---- S3Tokenizer --------------------------------------------------------------------------------------------------
@dataclass
class S3Config:
    """S3TokenizerModelConfig (S3TokenizerConfig.swift:9-50); V3 = 12 layers."""
    n_mels: int = 128
    n_audio_state: int = 1280
    n_audio_head: int = 20
    n_audio_layer: int = 6


S3_CONFIGS = {"s3_v2": S3Config(), "s3_v3": S3Config(n_audio_layer=12), "s3_micro": S3Config(128, 128, 2, 2)}


def s3_weights(cfg: S3Config, seed: int = 0) -> dict[str, np.ndarray]:
    """Random-init S3Tokenizer tensors with the reference's Module key paths (S3Tokenizer.swift)."""
    w: dict[str, np.ndarray] = {}
    D, M = cfg.n_audio_state, cfg.n_mels

    def t(name, shape, std):
        rng = np.random.Generator(np.random.PCG64(_key_seed(name, seed)))
        w[name] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

    def ln(p):
        t(p + ".weight", (D,), 0.1); w[p + ".weight"] += 1.0
        t(p + ".bias", (D,), 0.1)

    t("encoder.conv1.weight", (D, 3, M), 1.0 / math.sqrt(3 * M)); t("encoder.conv1.bias", (D,), 0.1)
    t("encoder.conv2.weight", (D, 3, D), 1.0 / math.sqrt(3 * D)); t("encoder.conv2.bias", (D,), 0.1)
    for l in range(cfg.n_audio_layer):
        p = f"encoder.blocks.{l}"
        ln(p + ".attn_ln"); ln(p + ".mlp_ln")
        for nm, bias in (("query", True), ("key", False), ("value", True), ("out", True)):
            t(f"{p}.attn.{nm}.weight", (D, D), 1.0 / math.sqrt(D))
            if bias:
                t(f"{p}.attn.{nm}.bias", (D,), 0.1)
        t(p + ".attn.fsmn_block.weight", (D, 31, 1), 1.0 / math.sqrt(31))
        t(p + ".mlp.layers.0.weight", (4 * D, D), 1.0 / math.sqrt(D)); t(p + ".mlp.layers.0.bias", (4 * D,), 0.1)
        t(p + ".mlp.layers.2.weight", (D, 4 * D), 1.0 / math.sqrt(4 * D)); t(p + ".mlp.layers.2.bias", (D,), 0.1)
    t("quantizer.fsq_codebook.project_down.weight", (8, D), 1.0 / math.sqrt(D))
    t("quantizer.fsq_codebook.project_down.bias", (8,), 0.1)
    return w


def qwen2lm_extra_weights(cfg: LMConfig, speech_token_size: int = 6561, seed: int = 0, round_to: str | None = None) -> dict[str, np.ndarray]:
    """The three CosyVoice2 tensors around the Qwen2 backbone (Qwen2LM.swift:259-263): llm_embedding [2,h], llm_decoder [S+3,h](+bias),
    speech_embedding [S+3,h]."""
    w = {}
    rng = np.random.Generator(np.random.PCG64(seed + 4242))
    h = cfg.hidden
    w["llm_embedding.weight"] = round_array(rng.standard_normal((2, h), dtype=np.float32) * np.float32(1.0 / math.sqrt(h) * 3.0), round_to)
    w["speech_embedding.weight"] = round_array(rng.standard_normal((speech_token_size + 3, h), dtype=np.float32) * np.float32(1.0 / math.sqrt(h) * 3.0), round_to)
    w["llm_decoder.weight"] = round_array(rng.standard_normal((speech_token_size + 3, h), dtype=np.float32) * np.float32(1.0 / math.sqrt(h) * 4.0), round_to)
    w["llm_decoder.bias"] = (0.1 * rng.standard_normal(speech_token_size + 3)).astype(np.float32)
    return w


# ---- HiFT vocoder (CosyHiFTGenerator.swift:283-300 defaults) ---------------------------------------------------------------
@dataclass(frozen=True)
class HiFTConfig:
    in_channels: int = 80
    base_channels: int = 512
    nb_harmonics: int = 8
    sampling_rate: int = 24000
    up_rates: tuple = (8, 5, 3)
    up_kernels: tuple = (16, 11, 7)
    res_kernels: tuple = (3, 7, 11)
    dilations: tuple = (1, 3, 5)
    src_res_kernels: tuple = (7, 7, 11)
    nsf_alpha: float = 0.1
    nsf_sigma: float = 0.003
    voiced_threshold: float = 10.0
    lrelu_slope: float = 0.1
    audio_limit: float = 0.99
    n_fft: int = 16
    hop: int = 4

    @property
    def upsample_factor(self) -> int:
        return int(np.prod(self.up_rates)) * self.hop


HIFT_CONFIGS = {"hift_cosyvoice2": HiFTConfig(), "hift_micro": HiFTConfig(base_channels=256)}


def hift_weights(cfg: HiFTConfig, seed: int = 0) -> dict[str, np.ndarray]:
    """Random-init CosyHiFTGenerator tensors with the reference's Module key paths (CosyHiFTGenerator.swift:272-279).  Scales are
    chosen so that the synthetic F0 spans voiced and unvoiced frames and the spectrum head stays inside exp()'s useful range."""
    w: dict[str, np.ndarray] = {}
    B, H = cfg.base_channels, cfg.nb_harmonics + 1

    def t(name, shape, std):
        rng = np.random.Generator(np.random.PCG64(_key_seed(name, seed)))
        w[name] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

    def conv(p, co, k, ci, gain=1.0):
        t(p + ".weight", (co, k, ci), gain / math.sqrt(k * ci)); t(p + ".bias", (co,), 0.05)

    def resblock(p, c, k):
        for i in range(len(cfg.dilations)):
            conv(f"{p}.convs1.{i}", c, k, c); conv(f"{p}.convs2.{i}", c, k, c, 0.4)
            for a in ("activations1", "activations2"):
                rng = np.random.Generator(np.random.PCG64(_key_seed(f"{p}.{a}.{i}.alpha", seed)))
                al = (1.0 + 0.5 * rng.standard_normal(c)).astype(np.float32)
                al[0] = 0.0; al[1] = 1e-6; al[2] = -1e-6          # exercise the clamp branches of Snake (HiFiGAN.swift:53-66)
                w[f"{p}.{a}.{i}.alpha"] = al

    for j, i in enumerate((0, 2, 4, 6, 8)):
        conv(f"f0_predictor.condnet_{i}", B, 3, cfg.in_channels if j == 0 else B, 1.3)
    t("f0_predictor.classifier.weight", (1, B), 70.0 / math.sqrt(B)); t("f0_predictor.classifier.bias", (1,), 1.0)
    t("m_source.l_linear.weight", (1, H), 1.0); t("m_source.l_linear.bias", (1,), 0.05)
    conv("conv_pre", B, 7, cfg.in_channels)
    n = len(cfg.up_rates)
    for i in range(n):
        ci, co = B >> i, B >> (i + 1)
        k, u = cfg.up_kernels[i], cfg.up_rates[i]
        t(f"ups.{i}.weight", (co, k, ci), 1.0 / math.sqrt(ci * k / u)); t(f"ups.{i}.bias", (co,), 0.05)
        dr = int(np.prod(cfg.up_rates[i + 1:])) if i + 1 < n else 1
        conv(f"source_downs.{i}", co, 1 if dr == 1 else 2 * dr, cfg.n_fft + 2, 0.5)
        resblock(f"source_resblocks.{i}", co, cfg.src_res_kernels[i])
        for k2, kk in enumerate(cfg.res_kernels):
            resblock(f"resblocks.{i * len(cfg.res_kernels) + k2}", co, kk)
    conv("conv_post", cfg.n_fft + 2, 7, B >> n, 0.25)
    return w


# ---- CosyVoice2 flow (FlowConfig, TTS/CosyVoice2/Config/CosyVoice2Config.swift:79-127) -----------------------------------------
@dataclass(frozen=True)
class FlowConfig:
    input_size: int = 512
    output_size: int = 80
    spk_embed_dim: int = 192
    vocab_size: int = 6561
    token_mel_ratio: int = 2
    pre_lookahead_len: int = 3
    n_timesteps: int = 10
    enc_heads: int = 8
    enc_linear_units: int = 2048
    enc_blocks: int = 6
    enc_up_blocks: int = 4
    upsample_stride: int = 2
    dec_in_channels: int = 320
    dec_channels: int = 256
    dec_heads: int = 8
    dec_n_blocks: int = 4
    dec_mid_blocks: int = 12
    cfg_rate: float = 0.7


FLOW_CONFIGS = {"flow_cosyvoice2": FlowConfig(),
                "flow_micro": FlowConfig(input_size=128, vocab_size=97, enc_heads=2, enc_linear_units=256, enc_blocks=2, enc_up_blocks=1,
                                         dec_channels=64, dec_heads=2, dec_n_blocks=1, dec_mid_blocks=2)}


def flow_weights(cfg: FlowConfig, seed: int = 0) -> dict[str, np.ndarray]:
    """Random-init CosyVoice2FlowModule tensors with the reference's (remapped) Module key paths (CosyVoice2TTS.swift:320-336)."""
    w: dict[str, np.ndarray] = {}
    D, M = cfg.input_size, cfg.output_size

    def t(name, shape, std):
        rng = np.random.Generator(np.random.PCG64(_key_seed(name, seed)))
        w[name] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

    def lin(p, o, i, bias=True, gain=1.0):
        t(p + ".weight", (o, i), gain / math.sqrt(i))
        if bias:
            t(p + ".bias", (o,), 0.05)

    def conv(p, o, k, i, gain=1.0):
        t(p + ".weight", (o, k, i), gain / math.sqrt(k * i)); t(p + ".bias", (o,), 0.05)

    def ln(p, d):
        t(p + ".weight", (d,), 0.1); w[p + ".weight"] += 1.0
        t(p + ".bias", (d,), 0.1)

    t("input_embedding.weight", (cfg.vocab_size, D), 1.0)
    lin("spk_embed_affine_layer", M, cfg.spk_embed_dim)
    e = "encoder"
    for p in (e + ".embed", e + ".up_embed"):
        lin(p + ".linear", D, D); ln(p + ".norm", D)
    conv(e + ".pre_lookahead_layer.conv1", D, cfg.pre_lookahead_len + 1, D)
    conv(e + ".pre_lookahead_layer.conv2", D, 3, D)
    conv(e + ".up_layer.conv", D, 2 * cfg.upsample_stride + 1, D)
    dk = D // cfg.enc_heads
    for grp, n in ((".encoders", cfg.enc_blocks), (".up_encoders", cfg.enc_up_blocks)):
        for i in range(n):
            p = f"{e}{grp}.{i}"
            for nm in ("linear_q", "linear_k", "linear_v", "linear_out"):
                lin(f"{p}.self_attn.{nm}", D, D, gain=0.7 if nm != "linear_out" else 0.5)
            lin(f"{p}.self_attn.linear_pos", D, D, bias=False)
            t(f"{p}.self_attn.pos_bias_u", (cfg.enc_heads, dk), 0.3); t(f"{p}.self_attn.pos_bias_v", (cfg.enc_heads, dk), 0.3)
            lin(f"{p}.feed_forward.w_1", cfg.enc_linear_units, D); lin(f"{p}.feed_forward.w_2", D, cfg.enc_linear_units, gain=0.5)
            ln(p + ".norm_ff", D); ln(p + ".norm_mha", D)
    ln(e + ".after_norm", D)
    lin("encoder_proj", M, D)
    d = "decoder.estimator"
    C, TE = cfg.dec_channels, 4 * cfg.dec_channels
    inner = cfg.dec_heads * 64
    lin(d + ".time_mlp.linear_1", TE, cfg.dec_in_channels); lin(d + ".time_mlp.linear_2", TE, TE)

    def resnet(p, ci, co):
        lin(p + ".mlp_linear", co, TE)
        conv(p + ".block1.conv.conv", co, 3, ci); ln(p + ".block1.norm", co)
        conv(p + ".block2.conv.conv", co, 3, co); ln(p + ".block2.norm", co)
        conv(p + ".res_conv", co, 1, ci)

    def tblock(p):
        ln(p + ".norm1", C); ln(p + ".norm3", C)
        for nm in ("query_proj", "key_proj", "value_proj"):
            lin(f"{p}.attn.{nm}", inner, C, bias=False)
        lin(p + ".attn.out_proj", C, inner, gain=0.5)
        lin(p + ".ff.layers.0", 4 * C, C); lin(p + ".ff.layers.1", C, 4 * C, gain=0.5)

    def block(p, ci):
        resnet(p + ".resnet", ci, C)
        for j in range(cfg.dec_n_blocks):
            tblock(f"{p}.transformers.{j}")

    block(d + ".down_blocks.0", cfg.dec_in_channels); conv(d + ".down_blocks.0.downsample.conv", C, 3, C)
    for i in range(cfg.dec_mid_blocks):
        block(f"{d}.mid_blocks.{i}", C)
    block(d + ".up_blocks.0", 2 * C); conv(d + ".up_blocks.0.upsample.conv", C, 3, C)
    conv(d + ".final_block.conv.conv", C, 3, C); ln(d + ".final_block.norm", C)
    conv(d + ".final_proj", M, 1, C)
    return w


# ---- CAM++ speaker encoder ----------------------------------------------------------------------------------------
CAMPP_BLOCKS = ((12, 3, 1), (24, 3, 2), (16, 3, 2))      # (layers, kernel, dilation) of the three dense blocks (CAMPPlus.swift:722)


def campplus_weights(seed: int = 0) -> dict[str, np.ndarray]:
    """Random-init CAMPPlus tensors with the reference's Module key paths (CAMPPlus.swift:687-753; the configuration
    CAMPlusSpeakerEncoder.swift:19-27 builds: 80 fbank bins, 192-d embedding, growth 32, bottleneck 128, 128 initial channels).
    MLX layouts: Conv2d [O, KH, KW, I], Conv1d [O, K, I].  He-scaled convolutions and BatchNorm statistics near (0, 1) keep
    every layer's activations O(1), so a wrong channel, tap or dilation shows up in the 192-d output."""
    w: dict[str, np.ndarray] = {}

    def rng_for(name):
        return np.random.Generator(np.random.PCG64(_key_seed(name, seed)))

    def conv(name, shape, bias=False):
        fan_in = int(np.prod(shape[1:]))
        w[name + ".weight"] = (rng_for(name + ".weight").standard_normal(shape, dtype=np.float32) * np.float32(math.sqrt(2.0 / fan_in)))
        if bias:
            w[name + ".bias"] = (0.1 * rng_for(name + ".bias").standard_normal(shape[0])).astype(np.float32)

    def bn(name, c, affine=True):
        r = rng_for(name)
        if affine:
            w[name + ".weight"] = (1.0 + 0.1 * r.standard_normal(c)).astype(np.float32)
            w[name + ".bias"] = (0.1 * r.standard_normal(c)).astype(np.float32)
        w[name + ".running_mean"] = (0.1 * r.standard_normal(c)).astype(np.float32)
        w[name + ".running_var"] = (0.5 + r.random(c)).astype(np.float32)

    m = 32
    conv("head.conv1", (m, 3, 3, 1)); bn("head.bn1", m)
    for layer in ("layer1", "layer2"):
        for i, stride in ((0, 2), (1, 1)):
            p = f"head.{layer}.{i}"
            conv(p + ".conv1", (m, 3, 3, m)); bn(p + ".bn1", m)
            conv(p + ".conv2", (m, 3, 3, m)); bn(p + ".bn2", m)
            if stride != 1:
                conv(p + ".shortcut.0", (m, 1, 1, m)); bn(p + ".shortcut.1", m)
    conv("head.conv2", (m, 3, 3, m)); bn("head.bn2", m)
    ch = m * (80 // 8)
    conv("tdnn.linear", (128, 5, ch)); bn("tdnn.nonlinear.0", 128)
    ch = 128
    for b, (n_layers, k, _dil) in enumerate(CAMPP_BLOCKS):
        for i in range(n_layers):
            p = f"blocks.{b}.layers.{i}"
            cin = ch + 32 * i
            bn(p + ".nonlinear1.0", cin)
            conv(p + ".linear1", (128, 1, cin))
            bn(p + ".nonlinear2.0", 128)
            conv(p + ".cam_layer.linear_local", (32, k, 128))
            conv(p + ".cam_layer.linear1", (64, 1, 128), bias=True)
            conv(p + ".cam_layer.linear2", (32, 1, 64), bias=True)
        ch += 32 * n_layers
        bn(f"transits.{b}.nonlinear.0", ch)
        conv(f"transits.{b}.linear", (ch // 2, 1, ch))
        ch //= 2
    bn("out_nonlinear.0", ch)
    conv("dense.linear", (192, 1, 2 * ch))
    bn("dense.nonlinear.0", 192, affine=False)
    return w
