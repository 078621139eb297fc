"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference's Llama-3 / Qwen2 decoder blocks and
the Orpheus sampler / frame parser (fp32, torch CPU).

parity unpinned: no golden vectors exist in the reference for this path.  MLX primitives restated from
the public documentation: RMSNorm(x) = x * rsqrt(mean(x^2) + eps) * w; RoPE(traditional: false) rotates the pairs (i, i + d/2)
by pos / freqs[i] (Llama3RoPE passes period-like `freqs`); scaledDotProductAttention repeats KV heads logically (GQA).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows (paths relative to /root/reference/package/TTS):
  Shared/Llama3RoPE.swift:27-66,104-114     Shared/SwiGLUMLP.swift:27-29
  Orpheus/BuildingBlocks/TransformerBlock.swift:70-105 (attention), :129-139 (block), :165-180 (model), :223-233 (tied lm head)
  CosyVoice2/LLM/Qwen2LM.swift:48-109 (attention with q/k/v bias), :113-151 (block)
  Orpheus/TTSEngine/OrpheusTTS.swift:375-470 (sampleNextToken), :472-508 (parseOutput)
"""
from __future__ import annotations

import math

import numpy as np
import torch

torch.set_grad_enabled(False)


def llama3_freqs(dims: int, base: float, llama3: bool, factor: float, low: float, high: float, old_ctx: float) -> np.ndarray:
    """Llama3RoPE.swift:41-65 in fp32: returns the period-like `freqs` MLX divides positions by."""
    f32 = np.float32
    idx = np.arange(0, dims, 2, dtype=f32)
    freqs = np.power(f32(base), idx / f32(dims)).astype(f32)
    if not llama3:
        return freqs
    low_wl, high_wl = f32(old_ctx) / f32(low), f32(old_ctx) / f32(high)
    wl = (f32(2.0) * f32(np.pi) * freqs).astype(f32)
    freqs = np.where(wl > low_wl, freqs * f32(factor), freqs).astype(f32)
    medium = (wl > high_wl) & (wl < low_wl)
    smooth = ((f32(old_ctx) / wl - f32(low)) / (f32(high) - f32(low))).astype(f32)
    smooth_freqs = (freqs / ((f32(1.0) - smooth) / f32(factor) + smooth)).astype(f32)
    return np.where(medium, smooth_freqs, freqs).astype(f32)


def _rms(x, w, eps):
    return x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + eps) * w


class LMOracle:
    def __init__(self, cfg, weights: dict[str, np.ndarray]):
        self.cfg = cfg
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v, np.float32)) for k, v in weights.items()}
        self.freqs = torch.from_numpy(llama3_freqs(cfg.head_dim, cfg.rope_theta, cfg.rope_llama3, cfg.rope_factor, cfg.rope_low,
                                                   cfg.rope_high, cfg.rope_old_ctx))
        self.reset()

    def reset(self):
        self.cache = [None] * self.cfg.n_layers
        self.offset = 0

    def _rope(self, x, offset):
        """x [H, L, d]: non-traditional (split-half) rotation."""
        L, d = x.shape[1], x.shape[2]
        pos = torch.arange(offset, offset + L, dtype=torch.float32)[:, None]
        ang = pos / self.freqs[None, :]
        cos, sin = torch.cos(ang), torch.sin(ang)
        x1, x2 = x[..., :d // 2], x[..., d // 2:]
        return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1)

    def forward(self, ids) -> torch.Tensor:
        """model(ids, cache) -> logits [L, vocab]; appends to the KV cache."""
        ids = torch.as_tensor(np.asarray(ids, np.int64))
        h = self.hidden(self.w["model.embed_tokens.weight"][ids])
        head = self.w["model.embed_tokens.weight"] if self.cfg.tie_embeddings else self.w["lm_head.weight"]
        return h @ head.t()

    def hidden(self, h: torch.Tensor) -> torch.Tensor:
        """Qwen2ModelInner.forward(embeddings:cache:) (Qwen2LM.swift:176-186): embeddings [L, hidden] -> final-norm hidden states."""
        c, W = self.cfg, self.w
        L = h.shape[0]
        for l in range(c.n_layers):
            p = f"model.layers.{l}"
            xn = _rms(h, W[p + ".input_layernorm.weight"], c.rms_eps)
            q = xn @ W[p + ".self_attn.q_proj.weight"].t()
            k = xn @ W[p + ".self_attn.k_proj.weight"].t()
            v = xn @ W[p + ".self_attn.v_proj.weight"].t()
            if c.qkv_bias:
                q = q + W[p + ".self_attn.q_proj.bias"]
                k = k + W[p + ".self_attn.k_proj.bias"]
                v = v + W[p + ".self_attn.v_proj.bias"]
            q = q.reshape(L, c.n_heads, c.head_dim).transpose(0, 1)
            k = k.reshape(L, c.n_kv_heads, c.head_dim).transpose(0, 1)
            v = v.reshape(L, c.n_kv_heads, c.head_dim).transpose(0, 1)
            q, k = self._rope(q, self.offset), self._rope(k, self.offset)
            if self.cache[l] is not None:
                k = torch.cat([self.cache[l][0], k], dim=1)
                v = torch.cat([self.cache[l][1], v], dim=1)
            self.cache[l] = (k, v)
            rep = c.n_heads // c.n_kv_heads
            kk, vv = k.repeat_interleave(rep, dim=0), v.repeat_interleave(rep, dim=0)
            s = (q @ kk.transpose(1, 2)) * (1.0 / math.sqrt(c.head_dim))
            if L > 1:                                                       # .causal for multi-token, .none for L == 1
                T = kk.shape[1]
                mask = torch.ones(L, T, dtype=torch.bool).tril(T - L)
                s = s.masked_fill(~mask, -float("inf"))
            o = (torch.softmax(s, dim=-1) @ vv).transpose(0, 1).reshape(L, -1)
            h = h + o @ W[p + ".self_attn.o_proj.weight"].t()
            xn = _rms(h, W[p + ".post_attention_layernorm.weight"], c.rms_eps)
            g = xn @ W[p + ".mlp.gate_proj.weight"].t()
            u = xn @ W[p + ".mlp.up_proj.weight"].t()
            h = h + (torch.nn.functional.silu(g) * u) @ W[p + ".mlp.down_proj.weight"].t()
        self.offset += L
        return _rms(h, W["model.norm.weight"], c.rms_eps)


def top_p_filter(logits: np.ndarray, history, rep_penalty: float, temperature: float, top_p: float) -> np.ndarray:
    """Steps 1-3 of sampleNextToken (OrpheusTTS.swift:388-461): returns the filtered logits (removed tokens = -inf)."""
    lg = torch.from_numpy(np.asarray(logits, np.float32)).clone()
    if rep_penalty != 1.0 and len(history) > 0:
        idx = torch.as_tensor(np.asarray(history, np.int64))
        g = lg[idx]
        lg[idx] = torch.where(g < 0, g * rep_penalty, g / rep_penalty)
    lg = lg / max(temperature, 1e-6)
    if 0.0 < top_p < 1.0 and lg.shape[0] > 1:
        probs = torch.softmax(lg, dim=-1)
        sidx = torch.argsort(-probs, stable=True)
        cum = torch.cumsum(probs[sidx], dim=-1)
        gt = (cum > top_p).to(torch.int32)
        remove_sorted = torch.cumsum(gt, dim=-1) > 1
        inv = torch.argsort(sidx)
        lg = torch.where(remove_sorted[inv], torch.tensor(-float("inf")), lg)
    return lg.numpy()


def sample_with_uniform(filtered_logits: np.ndarray, u: float) -> int:
    """The build's explicit-RNG categorical: inverse CDF over softmax(filtered) in index order (float64 cumsum)."""
    x = np.asarray(filtered_logits, np.float64)
    p = np.exp(x - x[np.isfinite(x)].max())
    p[~np.isfinite(x)] = 0.0
    c = np.cumsum(p)
    return int(np.searchsorted(c, u * c[-1], side="right"))


def sample_boundary_distance(logits: np.ndarray, history, rep_penalty: float, temperature: float, top_p: float, u: float) -> tuple[float, float]:
    """Oracle-only diagnostic for one sampler step: (distance of u to the nearest edge of the chosen token's CDF interval,
    distance of the nearest sorted cumulative probability to top_p).  A 16-bit build may legitimately pick another token only
    when one of the two is within its rounding noise."""
    f = top_p_filter(logits, history, rep_penalty, temperature, top_p)
    tok = sample_with_uniform(f, u)
    x = np.asarray(f, np.float64)
    p = np.exp(x - x[np.isfinite(x)].max())
    p[~np.isfinite(x)] = 0.0
    c = np.cumsum(p) / p.sum()
    lo = c[tok - 1] if tok > 0 else 0.0
    cdf_d = float(min(u - lo, c[tok] - u))
    lg = np.asarray(logits, np.float64).copy()
    if rep_penalty != 1.0 and len(history) > 0:
        idx = np.asarray(history, np.int64)
        lg[idx] = np.where(lg[idx] < 0, lg[idx] * rep_penalty, lg[idx] / rep_penalty)
    lg = lg / max(temperature, 1e-6)
    q = np.exp(lg - lg.max()); q /= q.sum()
    cum = np.cumsum(np.sort(q)[::-1])
    return cdf_d, float(np.abs(cum - top_p).min())


def generate(model: LMOracle, prompt, sampler, uniforms, trace: list | None = None) -> list[int]:
    """generateChunk's loop (OrpheusTTS.swift:245-348) with the explicit-uniform categorical.  `trace` (oracle-only) receives one
    sample_boundary_distance() pair per step."""
    model.reset()
    logits = model.forward(prompt)[-1].numpy()
    out, hist = [], []
    for i in range(sampler["max_new_tokens"]):
        f = top_p_filter(logits, hist, sampler["rep_penalty"], sampler["temperature"], sampler["top_p"])
        nxt = sample_with_uniform(f, float(uniforms[i]))
        if trace is not None:
            trace.append(sample_boundary_distance(logits, hist, sampler["rep_penalty"], sampler["temperature"], sampler["top_p"], float(uniforms[i])))
        out.append(nxt)
        if nxt in sampler["stop_ids"]:
            break
        hist.append(nxt)
        if len(hist) > sampler["rep_window"]:
            hist.pop(0)
        if i + 1 < sampler["max_new_tokens"]:
            logits = model.forward([nxt])[-1].numpy()
    return out


# Orpheus token constants (OrpheusTTS.swift:75-85)
END_TOKEN = 128258
CODE_OFFSET = 128266
AUDIO_CODE_DATA_START_MARKER = 128257


def parse_output(tokens: list[int]) -> list[list[int]]:
    """parseOutput (OrpheusTTS.swift:472-508): 7-token frames -> SNAC code lists (N, 2N, 4N)."""
    last = max((i for i, t in enumerate(tokens) if t == AUDIO_CODE_DATA_START_MARKER), default=-1)
    rel = tokens[last + 1:] if last >= 0 else tokens
    f = [t for t in rel if t != END_TOKEN and t >= CODE_OFFSET]
    f = [t - CODE_OFFSET for t in f[:(len(f) // 7) * 7]]
    l1, l2, l3 = [], [], []
    for i in range(len(f) // 7):
        b = 7 * i
        l1.append(f[b])
        l2.append(f[b + 1] - 4096)
        l3.append(f[b + 2] - 2 * 4096)
        l3.append(f[b + 3] - 3 * 4096)
        l2.append(f[b + 4] - 4 * 4096)
        l3.append(f[b + 5] - 5 * 4096)
        l3.append(f[b + 6] - 6 * 4096)
    return [l1, l2, l3]


# ---- CosyVoice2 Qwen2LM (TTS/CosyVoice2/LLM/Qwen2LM.swift:335-503) ---------------------------------------------------
def _inv_cdf(p: np.ndarray, u: float) -> int:
    c = np.cumsum(np.asarray(p, np.float64))
    return int(min(np.searchsorted(c, u * c[-1], side="right"), len(p) - 1))


def ras_sampling(logp: np.ndarray, decoded: list[int], draws, top_p=0.8, top_k=25, win=10, tau=0.1) -> int:
    """rasSampling (:463-488) with nucleusSampling (:433-461); `draws` yields the explicit uniforms."""
    x = np.asarray(logp, np.float64)
    probs = np.exp(x - x.max()); probs /= probs.sum()
    order = np.argsort(-probs, kind="stable")
    sp = probs[order]
    n = min(int((np.cumsum(sp) < top_p).sum()) + 1, top_k)
    tok = int(order[_inv_cdf(sp[:n], next(draws))])
    if decoded:
        rep = sum(1 for t in decoded[-win:] if t == tok)
        if rep >= win * tau:
            tok = _inv_cdf(probs, next(draws))
    return tok


def qwen2lm_inference(model: "LMOracle", lm_input: np.ndarray, head_w, head_b, speech_emb, eos: int, min_len: int, max_len: int, uniforms) -> list[int]:
    """inferenceLoop (:379-427): lm_input [n, hidden] embeddings; returns emitted speech tokens."""
    draws = iter(list(uniforms) + [uniforms[-1]] * 100000)
    hw, hb, se = torch.from_numpy(head_w), torch.from_numpy(head_b), torch.from_numpy(speech_emb)
    model.reset()
    out: list[int] = []
    cur = torch.from_numpy(np.ascontiguousarray(lm_input, np.float32))
    for i in range(max_len):
        y = model.hidden(cur)
        logits = y[-1] @ hw.t() + hb
        logp = torch.log(torch.softmax(logits, dim=-1)).numpy()
        trials = 0
        while True:
            top = ras_sampling(logp, out, draws)
            if not (i < min_len and top == eos):
                break
            trials += 1
            if trials > 100:
                break
        if top == eos:
            break
        cur = se[top][None, :]
        if top > eos:
            continue
        out.append(top)
    return out
