"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32) of CosyVoice2's token -> mel flow.

Follows, as text: CosyVoice2FlowModule.inference (TTS/CosyVoice2/CosyVoice2Model.swift:467-553), UpsampleConformerEncoder
(Codec/S3Gen/Transformer/UpsampleConformerEncoder.swift:14-102,407-474) with LinearNoSubsampling (Subsampling.swift:31-79),
RelPositionalEncoding (Embedding.swift:19-86), ConformerEncoderLayer (ConformerEncoderLayer.swift:69-165),
RelPositionMultiHeadedAttention (Attention.swift:119-196), PositionwiseFeedForward (PositionwiseFeedForward.swift:35-40);
CosyVoice2ConditionalCFM (TTS/CosyVoice2/Flow/CosyVoice2CFM.swift:74-187); ConditionalDecoder (Codec/S3Gen/S3GenDecoder.swift:
14-107,277-400) with SinusoidalPosEmb / TimestepEmbedding / mish (Matcha/MatchaDecoder.swift) and BasicTransformerBlock /
DiffusersAttention / FeedForward (Matcha/MatchaTransformer.swift:13-147).

Behaviour worth knowing (restated, not corrected): the encoder is built with RelPositionalEncoding (NOT the ESPnet two-sided
table), so pos_emb has T rows, matrix_bd already has the shape of matrix_ac and rel_shift never runs: the "relative" term is
an absolute-position key bias (q + v) . linear_pos(pe[j]).  The engine calls this path with batch 1 and full-length masks, so
every mask is all-ones; the restatement is written for that case.  The CFM's initial noise z is an explicit argument.

PARITY UNPINNED: the reference holds no golden vectors for this path and cannot run here; pinned by construction only.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _lin(w, p, x, bias=True):
    return F.linear(x, _t(w[p + ".weight"]), _t(w[p + ".bias"]) if bias and (p + ".bias") in w else None)


def _ln(w, p, x, eps):
    return F.layer_norm(x, (x.shape[-1],), _t(w[p + ".weight"]), _t(w[p + ".bias"]), eps)


def _conv(w, p, x_tc, stride=1, dilation=1):
    """MLX Conv1d, no padding, on [T, C] with weight [Cout, K, Cin] -> [T', Cout]."""
    y = F.conv1d(x_tc.T[None], _t(w[p + ".weight"]).permute(0, 2, 1), _t(w[p + ".bias"]), stride=stride, dilation=dilation)
    return y[0].T


def mish(x):
    return x * torch.tanh(torch.log(1 + torch.exp(x)))


def sinusoid_pe(T: int, d: int) -> torch.Tensor:
    """PositionalEncoding.createPE (Embedding.swift:33-52): interleaved sin / cos."""
    pos = torch.arange(T, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * np.float32(-math.log(10000.0) / d))
    pe = torch.stack([torch.sin(pos * div), torch.cos(pos * div)], dim=2).reshape(T, d)
    return pe


def chunk_mask(T: int, chunk: int) -> torch.Tensor | None:
    """subsequentChunkMask (UpsampleConformerEncoder.swift:124-129): [T, T] bool, query i sees keys j < (i // chunk + 1) * chunk; None = full."""
    if chunk <= 0:
        return None
    pos = torch.arange(T)
    return pos[None, :] < ((pos // chunk + 1) * chunk)[:, None]


def _masked_softmax(s, mask):
    """softmax over keys with where(mask, s, -inf) (Attention.swift:54-85; the estimator adds the same mask as a -inf bias)."""
    if mask is not None:
        s = s.masked_fill(~mask[None], -float("inf"))
    return torch.softmax(s, dim=-1)


def rel_attention(w, p, x, pos_emb, H, mask=None):
    """RelPositionMultiHeadedAttention.callAsFunction (Attention.swift:143-195), no cache; mask [T, T] bool or None (all ones)."""
    T, D = x.shape
    dk = D // H
    q = _lin(w, p + ".linear_q", x).reshape(T, H, dk)
    k = _lin(w, p + ".linear_k", x).reshape(T, H, dk).permute(1, 0, 2)
    v = _lin(w, p + ".linear_v", x).reshape(T, H, dk).permute(1, 0, 2)
    pp = F.linear(pos_emb, _t(w[p + ".linear_pos.weight"])).reshape(-1, H, dk).permute(1, 0, 2)
    qu = (q + _t(w[p + ".pos_bias_u"])).permute(1, 0, 2)
    qv = (q + _t(w[p + ".pos_bias_v"])).permute(1, 0, 2)
    ac = qu @ k.transpose(1, 2)
    bd = qv @ pp.transpose(1, 2)
    assert ac.shape == bd.shape          # -> relShift is skipped (Attention.swift:186-188)
    att = _masked_softmax((ac + bd) / math.sqrt(dk), mask)
    o = (att @ v).permute(1, 0, 2).reshape(T, D)
    return _lin(w, p + ".linear_out", o)


def conformer_layer(w, p, x, pos_emb, H, mask=None):
    x = x + rel_attention(w, p + ".self_attn", _ln(w, p + ".norm_mha", x, 1e-12), pos_emb, H, mask)
    h = _ln(w, p + ".norm_ff", x, 1e-12)
    h = _lin(w, p + ".feed_forward.w_2", F.silu(_lin(w, p + ".feed_forward.w_1", h)))
    return x + h


def embed(w, p, x, D):
    """LinearNoSubsampling + RelPositionalEncoding: (x W + b) -> LayerNorm(1e-5) -> * sqrt(D); pos_emb = pe[0:T]."""
    x = _ln(w, p + ".norm", _lin(w, p + ".linear", x), 1e-5) * np.float32(math.sqrt(D))
    return x, sinusoid_pe(x.shape[0], D)


def encoder(w, cfg, x, static_chunk: int = 0):
    """UpsampleConformerEncoder.callAsFunction (:407-474); static_chunk > 0 = streaming: chunk masks of that size, and of
    static_chunk * upsample_stride for the up-sampled blocks (:424-460)."""
    D, H = cfg.input_size, cfg.enc_heads
    p = "encoder"
    x, pe = embed(w, p + ".embed", x, D)
    # PreLookaheadLayer (:86-101)
    L = cfg.pre_lookahead_len
    h = F.pad(x, (0, 0, 0, L))
    h = F.leaky_relu(_conv(w, p + ".pre_lookahead_layer.conv1", h), 0.01)
    h = F.pad(h, (0, 0, 2, 0))
    x = _conv(w, p + ".pre_lookahead_layer.conv2", h) + x
    for i in range(cfg.enc_blocks):
        x = conformer_layer(w, f"{p}.encoders.{i}", x, pe, H, chunk_mask(x.shape[0], static_chunk))
    # Upsample1D (:36-55): repeat, left pad 2 * stride, conv k = 2 * stride + 1
    s = cfg.upsample_stride
    x = torch.repeat_interleave(x, s, dim=0)
    x = F.pad(x, (0, 0, 2 * s, 0))
    x = _conv(w, p + ".up_layer.conv", x)
    x, pe = embed(w, p + ".up_embed", x, D)
    for i in range(cfg.enc_up_blocks):
        x = conformer_layer(w, f"{p}.up_encoders.{i}", x, pe, H, chunk_mask(x.shape[0], static_chunk * cfg.upsample_stride))
    return _ln(w, p + ".after_norm", x, 1e-5)


# ---- estimator ----------------------------------------------------------------------------------------------------------------
def causal_block(w, p, x):
    """CausalBlock1D (:62-70) on [T, C], mask all ones."""
    h = _conv(w, p + ".conv.conv", F.pad(x, (0, 0, 2, 0)))
    return mish(_ln(w, p + ".norm", h, 1e-5))


def resnet(w, p, x, temb):
    h = causal_block(w, p + ".block1", x)
    h = h + _lin(w, p + ".mlp_linear", mish(temb))[None, :]
    h = causal_block(w, p + ".block2", h)
    return h + _conv(w, p + ".res_conv", x)


def transformer(w, p, x, H, mask=None):
    T, _ = x.shape
    n = _ln(w, p + ".norm1", x, 1e-5)
    q = F.linear(n, _t(w[p + ".attn.query_proj.weight"])).reshape(T, H, 64).permute(1, 0, 2)
    k = F.linear(n, _t(w[p + ".attn.key_proj.weight"])).reshape(T, H, 64).permute(1, 0, 2)
    v = F.linear(n, _t(w[p + ".attn.value_proj.weight"])).reshape(T, H, 64).permute(1, 0, 2)
    att = _masked_softmax((q @ k.transpose(1, 2)) * (64 ** -0.5), mask)
    o = (att @ v).permute(1, 0, 2).reshape(T, H * 64)
    x = x + _lin(w, p + ".attn.out_proj", o)
    n = _ln(w, p + ".norm3", x, 1e-5)
    return x + _lin(w, p + ".ff.layers.1", F.gelu(_lin(w, p + ".ff.layers.0", n)))


def time_embedding(w, cfg, t: float) -> torch.Tensor:
    """SinusoidalPosEmb(dim = in_channels, scale 1000) -> TimestepEmbedding (MatchaDecoder.swift:13-58)."""
    half = cfg.dec_in_channels // 2
    emb = torch.exp(torch.arange(half, dtype=torch.float32) * -np.float32(math.log(10000.0) / (half - 1)))
    arg = np.float32(1000.0) * np.float32(t) * emb
    e = torch.cat([torch.sin(arg), torch.cos(arg)])
    p = "decoder.estimator.time_mlp"
    return _lin(w, p + ".linear_2", F.silu(_lin(w, p + ".linear_1", e)))


def estimator(w, cfg, x, mu, t, spks, cond, static_chunk: int = 0):
    """ConditionalDecoder.callAsFunction (:277-400) for one batch element; x, mu, cond [T, 80]; spks [80]; static_chunk > 0 = the
    streaming attention bias of :304-320."""
    p = "decoder.estimator"
    H = cfg.dec_heads
    mask = chunk_mask(x.shape[0], static_chunk)
    temb = time_embedding(w, cfg, t)
    h = torch.cat([x, mu, spks[None, :].expand(x.shape[0], -1), cond], dim=1)
    h = resnet(w, p + ".down_blocks.0.resnet", h, temb)
    for j in range(cfg.dec_n_blocks):
        h = transformer(w, f"{p}.down_blocks.0.transformers.{j}", h, H, mask)
    skip = h
    h = _conv(w, p + ".down_blocks.0.downsample.conv", F.pad(h, (0, 0, 2, 0)))
    for i in range(cfg.dec_mid_blocks):
        h = resnet(w, f"{p}.mid_blocks.{i}.resnet", h, temb)
        for j in range(cfg.dec_n_blocks):
            h = transformer(w, f"{p}.mid_blocks.{i}.transformers.{j}", h, H, mask)
    h = torch.cat([h, skip], dim=1)
    h = resnet(w, p + ".up_blocks.0.resnet", h, temb)
    for j in range(cfg.dec_n_blocks):
        h = transformer(w, f"{p}.up_blocks.0.transformers.{j}", h, H, mask)
    h = _conv(w, p + ".up_blocks.0.upsample.conv", F.pad(h, (0, 0, 2, 0)))
    h = causal_block(w, p + ".final_block", h)
    return _conv(w, p + ".final_proj", h)


def t_span(n: int) -> np.ndarray:
    t = np.linspace(np.float32(0), np.float32(1), n + 1, dtype=np.float32)
    return (np.float32(1) - np.cos(t * np.float32(0.5) * np.float32(np.pi))).astype(np.float32)


def cfm(w, cfg, mu, spks, cond, z, n_timesteps, static_chunk: int = 0):
    """CosyVoice2ConditionalCFM: cosine schedule + solveEuler with classifier-free guidance (:74-187)."""
    ts = t_span(n_timesteps)
    x = z.clone()
    t = ts[0]
    dt = ts[1] - ts[0]
    zmu, zspk, zcond = torch.zeros_like(mu), torch.zeros_like(spks), torch.zeros_like(cond)
    rate = np.float32(cfg.cfg_rate)
    for step in range(1, n_timesteps + 1):
        d_c = estimator(w, cfg, x, mu, float(t), spks, cond, static_chunk)
        d_u = estimator(w, cfg, x, zmu, float(t), zspk, zcond, static_chunk)
        x = x + float(dt) * ((1.0 + float(rate)) * d_c - float(rate) * d_u)
        t = np.float32(t + dt)
        if step < n_timesteps:
            dt = np.float32(ts[step + 1] - t)
    return x


def inference(w, cfg, token, prompt_token, prompt_feat, embedding, z, n_timesteps=None, finalize=True, enc_static_chunk=0, dec_static_chunk=0):
    """CosyVoice2FlowModule.inference (CosyVoice2Model.swift:467-553).  token [n], prompt_token [m] int; prompt_feat [2 m, 80]; embedding
    [192]; z [80, T] -> mel [80, T - 2 m], T = 2 (n + m), minus pre_lookahead_len * 2 frames when finalize is false (:504-510).
    enc_static_chunk / dec_static_chunk > 0: the modules' streaming masks (the engine itself always passes streaming: false)."""
    n_timesteps = n_timesteps or cfg.n_timesteps
    emb = _t(embedding)
    emb = emb / (torch.sqrt((emb * emb).sum()) + 1e-8)
    spks = _lin(w, "spk_embed_affine_layer", emb)
    full = np.concatenate([prompt_token, token]).astype(np.int64)
    full = np.clip(full, 0, w["input_embedding.weight"].shape[0] - 1)
    x = _t(w["input_embedding.weight"])[torch.from_numpy(full)]
    enc = encoder(w, cfg, x, enc_static_chunk)
    if not finalize:
        trim = cfg.pre_lookahead_len * cfg.token_mel_ratio
        if enc.shape[0] > trim:
            enc = enc[:enc.shape[0] - trim]
    mu = _lin(w, "encoder_proj", enc)
    T = mu.shape[0]
    m1 = prompt_feat.shape[0]
    cond = torch.zeros(T, cfg.output_size)
    cond[:m1] = _t(prompt_feat)
    mel = cfm(w, cfg, mu, spks, cond, _t(z).T.contiguous(), n_timesteps, dec_static_chunk)
    return mel[m1:].T.contiguous().numpy(), {"mu": mu.numpy(), "spks": spks.numpy(), "enc": enc.numpy()}
