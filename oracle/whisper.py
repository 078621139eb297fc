"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference Whisper network and greedy decoder.

parity unpinned: the reference holds no tensor-level golden vectors for this path; this file
restates the Swift source line by line in fp32 (torch CPU for the matmuls) and is cross-checked against the
independent `transformers` Whisper implementation in tests/test_oracle_whisper.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows (paths relative to /root/reference/package/STT/Whisper):
  Layers/AudioEncoder.swift:43-68,78-96        encoder forward, sinusoids
  Layers/MultiHeadAttention.swift:40-135       q/k/v projections (key has no bias), d^-1/4 scaling of q and k,
                                               additive mask slice, fp32 softmax, output projection
  Layers/ResidualAttentionBlock.swift:51-95    pre-LN block order
  Layers/TextDecoder.swift:38-42,53-96         causal -inf mask, learned positions by cache offset, tied logits
  WhisperDecoding.swift:96-389                 GreedyDecoder.decode (initial tokens, rules, argmax, log-probs)
  WhisperModel.swift:121-128,223-260           vocabulary arithmetic, detectLanguage
  WhisperTokenizer.swift:72-96,377-396         special-token ids, sotSequence
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch

from mlx_swift_audio_amd.synthetic import (DIMS, ModelDimensions, round_array, synthetic_suppress_list,  # noqa: F401
                                           synthetic_weights, weight_names)

torch.set_grad_enabled(False)


@dataclass
class SpecialTokens:
    """WhisperTokenizer.swift:72-96 id arithmetic."""
    eot: int
    sot: int
    translate: int
    transcribe: int
    sot_lm: int
    sot_prev: int
    no_speech: int
    no_timestamps: int
    timestamp_begin: int
    is_multilingual: bool
    num_languages: int

    @staticmethod
    def for_vocab(n_vocab: int) -> "SpecialTokens":
        multilingual = n_vocab >= 51865                      # WhisperModel.swift:121-123
        num_languages = n_vocab - 51765 - (1 if multilingual else 0)   # :126-128
        nxt = 50257 if multilingual else 50256
        eot = nxt; nxt += 1
        sot = nxt; nxt += 1
        nxt += num_languages
        translate = nxt; nxt += 1
        transcribe = nxt; nxt += 1
        sot_lm = nxt; nxt += 1
        sot_prev = nxt; nxt += 1
        no_speech = nxt; nxt += 1
        no_timestamps = nxt; nxt += 1
        return SpecialTokens(eot, sot, translate, transcribe, sot_lm, sot_prev, no_speech, no_timestamps, nxt,
                             multilingual, num_languages)

    def sot_sequence(self, language_index: int | None = 0, task: str = "transcribe") -> list[int]:
        """WhisperTokenizer.swift:377-396 (language token = sot + 1 + index)."""
        seq = [self.sot]
        if not self.is_multilingual:
            return seq
        if language_index is not None:
            seq.append(self.sot + 1 + language_index)
        seq.append(self.transcribe if task == "transcribe" else self.translate)
        return seq


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> np.ndarray:
    """AudioEncoder.swift:78-96 in fp32."""
    inc = np.float32(math.log(max_timescale)) / np.float32(channels // 2 - 1)
    inv = np.exp(-inc * np.arange(channels // 2, dtype=np.float32)).astype(np.float32)
    st = (np.arange(length, dtype=np.float32)[:, None] * inv[None, :]).astype(np.float32)
    return np.concatenate([np.sin(st), np.cos(st)], axis=1).astype(np.float32)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32))


def _linear(x, w, b=None):
    y = x @ w.t()
    return y + b if b is not None else y


def _layer_norm(x, g, b, eps=1e-5):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), g, b, eps)


def _gelu(x):
    return torch.nn.functional.gelu(x)      # exact erf form == MLXNN GELU()


def _conv1d_cl(x, w, b, stride):
    """MLX Conv1d: input [B,L,Cin] channels-last, weight [Cout,K,Cin], padding 1."""
    y = torch.nn.functional.conv1d(x.transpose(1, 2), w.permute(0, 2, 1), b, stride=stride, padding=1)
    return y.transpose(1, 2)


class WhisperOracle:
    """fp32 restatement of WhisperModel (encoder + decoder)."""

    def __init__(self, dims: ModelDimensions, weights: dict[str, np.ndarray], threads: int | None = None):
        self.dims = dims
        self.w = {k: _t(v) for k, v in weights.items()}
        if "encoder.positional_embedding" not in self.w:
            self.w["encoder.positional_embedding"] = _t(sinusoids(dims.n_audio_ctx, dims.n_audio_state))
        if threads:
            torch.set_num_threads(threads)
        n = dims.n_text_ctx
        idx = torch.arange(n)
        self.mask = torch.where(idx[:, None] < idx[None, :], torch.tensor(-float("inf")), torch.tensor(0.0))  # TextDecoder.swift:38-42

    # ---- MultiHeadAttention.swift:85-135
    def _qkv_attention(self, q, k, v, n_head, mask=None, offset=0):
        B, n_ctx, n_state = q.shape
        scale = float((n_state // n_head)) ** -0.25
        qh = q.reshape(B, n_ctx, n_head, -1).permute(0, 2, 1, 3) * scale
        k_ctx = k.shape[1]
        kh = k.reshape(B, k_ctx, n_head, -1).permute(0, 2, 3, 1) * scale
        vh = v.reshape(B, k_ctx, n_head, -1).permute(0, 2, 1, 3)
        qk = qh @ kh
        if mask is not None:
            qk = qk + mask[offset:offset + n_ctx, :k_ctx]
        wts = torch.softmax(qk, dim=-1)
        out = (wts @ vh).permute(0, 2, 1, 3).reshape(B, n_ctx, n_state)
        return out, qk

    def _attn(self, p, x, n_head, xa=None, mask=None, kv_cache=None, offset=0):
        W = self.w
        q = _linear(x, W[p + ".query.weight"], W[p + ".query.bias"])
        if xa is not None:
            if kv_cache is not None:
                k, v = kv_cache
            else:
                k = _linear(xa, W[p + ".key.weight"])
                v = _linear(xa, W[p + ".value.weight"], W[p + ".value.bias"])
        else:
            k = _linear(x, W[p + ".key.weight"])
            v = _linear(x, W[p + ".value.weight"], W[p + ".value.bias"])
            if kv_cache is not None:
                k = torch.cat([kv_cache[0], k], dim=1)
                v = torch.cat([kv_cache[1], v], dim=1)
        wv, qk = self._qkv_attention(q, k, v, n_head, mask, offset)
        return _linear(wv, W[p + ".out.weight"], W[p + ".out.bias"]), (k, v), qk

    # ---- ResidualAttentionBlock.swift:51-95
    def _block(self, p, x, n_head, xa=None, mask=None, kv_cache=None, offset=0, cross_qk_out=None):
        W = self.w
        self_kv = kv_cache[0] if kv_cache else None
        cross_kv = kv_cache[1] if kv_cache else None
        y, new_self, _ = self._attn(p + ".attn", _layer_norm(x, W[p + ".attn_ln.weight"], W[p + ".attn_ln.bias"]), n_head,
                                    mask=mask, kv_cache=self_kv, offset=offset)
        x = x + y
        new_cross = cross_kv
        if xa is not None:
            y, new_cross, cqk = self._attn(p + ".cross_attn", _layer_norm(x, W[p + ".cross_attn_ln.weight"], W[p + ".cross_attn_ln.bias"]),
                                           n_head, xa=xa, kv_cache=cross_kv)
            if cross_qk_out is not None:
                cross_qk_out.append(cqk)
            x = x + y
        h = _layer_norm(x, W[p + ".mlp_ln.weight"], W[p + ".mlp_ln.bias"])
        x = x + _linear(_gelu(_linear(h, W[p + ".mlp1.weight"], W[p + ".mlp1.bias"])), W[p + ".mlp2.weight"], W[p + ".mlp2.bias"])
        return x, (new_self, new_cross)

    # ---- AudioEncoder.swift:43-68
    def encode(self, mel: np.ndarray) -> torch.Tensor:
        """mel [B, 2*n_audio_ctx, n_mels] -> [B, n_audio_ctx, n_audio_state]."""
        W, d = self.w, self.dims
        x = _t(mel)
        x = _gelu(_conv1d_cl(x, W["encoder.conv1.weight"], W["encoder.conv1.bias"], 1))
        x = _gelu(_conv1d_cl(x, W["encoder.conv2.weight"], W["encoder.conv2.bias"], 2))
        x = x + W["encoder.positional_embedding"][:x.shape[1]]
        for l in range(d.n_audio_layer):
            x, _ = self._block(f"encoder.blocks.{l}", x, d.n_audio_head)
        return _layer_norm(x, W["encoder.ln_post.weight"], W["encoder.ln_post.bias"])

    # ---- TextDecoder.swift:53-96
    def decode(self, tokens: list[int] | np.ndarray, xa: torch.Tensor, kv_cache=None, cross_qk_out=None):
        W, d = self.w, self.dims
        tok = torch.as_tensor(np.asarray(tokens, np.int64)).reshape(1, -1)
        offset = kv_cache[0][0][0].shape[1] if kv_cache is not None and kv_cache[0][0] is not None else 0
        n = tok.shape[-1]
        x = W["decoder.token_embedding.weight"][tok] + W["decoder.positional_embedding"][offset:offset + n]
        new_cache = []
        for l in range(d.n_text_layer):
            x, c = self._block(f"decoder.blocks.{l}", x, d.n_text_head, xa=xa, mask=self.mask,
                               kv_cache=kv_cache[l] if kv_cache is not None else None, offset=offset, cross_qk_out=cross_qk_out)
            new_cache.append(c)
        x = _layer_norm(x, W["decoder.ln.weight"], W["decoder.ln.bias"])
        return x @ W["decoder.token_embedding.weight"].t(), new_cache

    # ---- WhisperModel.swift:223-260
    def detect_language(self, xa: torch.Tensor, st: SpecialTokens):
        logits, _ = self.decode([st.sot], xa)
        ll = logits[0, 0, st.sot + 1: st.sot + 1 + st.num_languages]
        probs = torch.softmax(ll, dim=-1)
        i = int(torch.argmax(probs))
        return i, float(probs[i])


@dataclass
class DecodingOptions:
    """WhisperDecoding.swift:14-51 (language as an index; tokenizer outputs passed as integer tables)."""
    task: str = "transcribe"
    language_index: int | None = 0
    temperature: float = 0.0
    max_tokens: int = 448
    timestamps: bool = True
    prompt: list[int] = field(default_factory=list)
    suppress_ids: list[int] = field(default_factory=list)   # nonSpeechTokens + specials (:190-198)
    blank_ids: list[int] = field(default_factory=list)      # tokenizer.encode(" ") (:201-206)
    max_new_tokens: int = 0                                 # benchmarking aid (not in the reference)
    max_initial_timestamp_index: int = 50


@dataclass
class DecodingResult:
    tokens: list[int]
    avg_logprob: float
    no_speech_prob: float
    margins: list[float]          # oracle-only diagnostic: top-1 minus top-2 filtered logit at every step
    initial_tokens: list[int]
    heur: list = None             # oracle-only diagnostic: ts_logprob - max_text_logprob of the raw-logit timestamp heuristic (None where
                                  # it does not run): the decision flips when this crosses 0
    cdf_margins: list = None      # oracle-only diagnostic (T > 0): distance of the step's uniform to the nearest edge of the chosen
                                  # token's CDF interval, i.e. how far the 16-bit build's CDF may move before the draw changes


def initial_tokens(st: SpecialTokens, o: DecodingOptions) -> tuple[list[int], int]:
    """WhisperDecoding.swift:104-122.  Returns (tokens, sot_index)."""
    toks: list[int] = []
    if o.prompt:
        toks.append(st.sot_prev)
        toks.extend(o.prompt)
    sot_index = len(toks)
    toks.extend(st.sot_sequence(o.language_index, o.task))
    if not o.timestamps:
        toks.append(st.no_timestamps)
    return toks, sot_index


def sample_from_distribution(probs: np.ndarray, r: float) -> int:
    """sampleFromDistribution (WhisperDecoding.swift:395-410): sequential fp32 cumsum, first index with cumsum >= r."""
    c = np.cumsum(probs.astype(np.float32), dtype=np.float32)
    i = int(np.searchsorted(c, np.float32(r), side="left"))
    return min(i, probs.shape[0] - 1)


def rule_masks(tokens: list[int], initial_count: int, st: SpecialTokens, o: DecodingOptions, V: int):
    """The two additive masks (0 / -inf) GreedyDecoder builds before the timestamp-probability heuristic: the suppress-token mask
    (WhisperDecoding.swift:190-206: the caller's list, plus blanks and EOT before the first generated token) and the timestamp-rule mask
    (:214-290: <|notimestamps|>, timestamps in pairs, monotonic timestamps with the port's strict `> timestampBegin` filter and its
    `+1 iff penultimateWasTimestamp`, first token a timestamp no later than max_initial_timestamp).  tests/test_oracle_whisper_rules.py
    compares them with `transformers`' WhisperTimeStampLogitsProcessor / SuppressTokens processors."""
    NEG = -float("inf")
    idx = torch.arange(V)
    num_generated = len(tokens) - initial_count
    sup = list(o.suppress_ids)                                           # :190-198 (caller passes the full list)
    if num_generated == 0:
        sup = sup + list(o.blank_ids) + [st.eot]                         # :201-206
    base = torch.zeros(V)
    for t in sup:
        if t < V:
            base[t] = NEG
    ts_mask = torch.zeros(V)
    tsb = st.timestamp_begin
    if o.timestamps:
        ts_mask[st.no_timestamps] = NEG
        last_was_ts = num_generated >= 1 and tokens[-1] >= tsb
        penult_was_ts = num_generated < 2 or tokens[-2] >= tsb
        if last_was_ts:
            if penult_was_ts:
                ts_mask[idx >= tsb] = NEG
            else:
                ts_mask[idx < st.eot] = NEG
        gen = tokens[len(tokens) - num_generated:] if num_generated else []
        ts_vals = [t for t in gen if t > tsb]                            # strict > (:254-256)
        if ts_vals:
            lt = ts_vals[-1] + (1 if penult_was_ts else 0)
            ts_mask[(idx >= tsb) & (idx < lt)] = NEG
        if num_generated == 0:
            ts_mask[idx < tsb] = NEG
            last_allowed = tsb + o.max_initial_timestamp_index
            if last_allowed < V:
                ts_mask[idx > last_allowed] = NEG
    return base, ts_mask


def filter_logits(last: torch.Tensor, tokens: list[int], initial_count: int, st: SpecialTokens, o: DecodingOptions):
    """One iteration's logit rules (WhisperDecoding.swift:186-330) on the raw logits `last` [V] of the position that predicts the next
    token, given the sequence so far: suppress / timestamp-rule masks, then the timestamp-probability heuristic evaluated on the RAW
    logits (:299-322).  Returns (filtered logits, heuristic distance) -- the distance ts_logprob - max_text_logprob is what the
    heuristic thresholds at 0 (None when the heuristic does not run)."""
    V = last.shape[-1]
    NEG = -float("inf")
    idx = torch.arange(V)
    num_generated = len(tokens) - initial_count
    base, ts_mask = rule_masks(tokens, initial_count, st, o, V)
    tsb = st.timestamp_begin
    dist = None
    if o.timestamps and num_generated > 0:                           # :299-322 on RAW logits
        lp = last - torch.logsumexp(last, dim=-1, keepdim=True)
        ts_lp = torch.logsumexp(lp[tsb:], dim=-1)
        max_text = lp[:tsb].max()
        dist = float(ts_lp) - float(max_text)
        if float(ts_lp) > float(max_text):
            ts_mask[idx < tsb] = NEG
    return last + torch.minimum(base, ts_mask), dist


def greedy_decode(model: WhisperOracle, st: SpecialTokens, xa: torch.Tensor, o: DecodingOptions, uniforms=None) -> DecodingResult:
    """GreedyDecoder.decode (WhisperDecoding.swift:96-389) for ONE clip (xa [1, n_audio_ctx, D]).  temperature 0 = argmax;
    temperature > 0 samples with `uniforms[k]` standing in for the k-th Float.random(in: 0..<1) of the reference."""
    assert o.temperature == 0.0 or uniforms is not None, "T>0 needs explicit uniforms (the reference's RNG is unseeded)"
    tokens, sot_index = initial_tokens(st, o)
    init = list(tokens)
    initial_count = len(tokens)
    max_generate = o.max_tokens - initial_count
    if o.max_new_tokens > 0:
        max_generate = min(max_generate, o.max_new_tokens)
    kv = None
    sum_lp, count, no_speech_prob = 0.0, 0, 0.0
    margins: list[float] = []
    cdf_margins: list[float] = []
    heur: list = []
    for it in range(max_generate):
        feed = tokens if kv is None else tokens[-1:]
        logits, kv = model.decode(feed, xa, kv)
        if it == 0:
            probs = torch.softmax(logits[0, sot_index], dim=-1)          # :158-169
            no_speech_prob = float(probs[st.no_speech])
        last, hd = filter_logits(logits[0, -1].clone(), tokens, initial_count, st, o)
        heur.append(hd)
        cdf_m = float("inf")
        if o.temperature == 0.0:
            nxt = int(torch.argmax(last))
        elif not bool(torch.isfinite(last).any()):
            nxt = 0
        else:
            probs = torch.softmax(last / o.temperature, dim=-1).numpy()            # :335-338
            nxt = sample_from_distribution(probs, float(uniforms[it]))
            c = np.cumsum(probs.astype(np.float64))
            lo = c[nxt - 1] if nxt > 0 else 0.0
            cdf_m = float(min(float(uniforms[it]) - lo, c[nxt] - float(uniforms[it])))
        cdf_margins.append(cdf_m)
        top2 = torch.topk(last, 2).values
        margins.append(float(top2[0] - top2[1]))
        if nxt != st.eot:                                                # :345-350
            lps = torch.log(torch.softmax(last, dim=-1))
            sum_lp += float(lps[nxt])
            count += 1
        tokens.append(nxt)
        if nxt == st.eot:
            break
    avg = sum_lp / count if count > 0 else 0.0
    gen = tokens[initial_count:]
    if st.eot in gen:
        gen = gen[:gen.index(st.eot)]
    return DecodingResult(gen, avg, no_speech_prob, margins, init, heur, cdf_margins)


def teacher_forced_logits(model: WhisperOracle, xa: torch.Tensor, tokens: list[int]) -> np.ndarray:
    """Raw logits [len(tokens), V] of ONE causal pass over `tokens` (TextDecoder.swift:53-96 without a cache): row p is what the
    decoder predicts after consuming tokens[0..p] -- the quantity a step-by-step decoder with a KV cache computes at position p."""
    logits, _ = model.decode(list(tokens), xa)
    return logits[0].numpy()


def replay_rules(step_logits: np.ndarray, tokens: list[int], initial_count: int, st: SpecialTokens, o: DecodingOptions):
    """Run the decode head's decision logic (filter_logits + argmax + log-prob bookkeeping, WhisperDecoding.swift:186-350) on GIVEN
    per-position raw logits (row p predicts tokens[p + 1]) along the GIVEN token sequence: what a decoder whose logits these are must
    have emitted.  Returns (ids per generated position, margins, heuristic distances, avg_logprob)."""
    ids, margins, dists = [], [], []
    sum_lp, count = 0.0, 0
    for p in range(initial_count - 1, len(tokens) - 1):
        last, dist = filter_logits(torch.from_numpy(np.ascontiguousarray(step_logits[p], np.float32)).clone(), list(tokens[:p + 1]), initial_count, st, o)
        nxt = int(torch.argmax(last))
        top2 = torch.topk(last, 2).values
        margins.append(float(top2[0] - top2[1]))
        dists.append(dist)
        ids.append(nxt)
        emitted = tokens[p + 1]
        if emitted != st.eot:
            sum_lp += float(torch.log(torch.softmax(last, dim=-1))[emitted])
            count += 1
    return ids, margins, dists, (sum_lp / count if count else 0.0)


# ---- word timestamps: WhisperTiming.swift (dtw :46-130, medianFilterAttention :191-253, findAlignment :558-748) ----------------------
def dtw(cost: np.ndarray):
    """cost [N, M] -> (text_indices, time_indices), tie rules of WhisperTiming.swift:70-78."""
    N, M = cost.shape
    acc = np.full((N + 1, M + 1), np.inf, np.float32)
    trace = -np.ones((N + 1, M + 1), np.int8)
    acc[0, 0] = 0
    for j in range(1, M + 1):
        for i in range(1, N + 1):
            c0, c1, c2 = acc[i - 1, j - 1], acc[i - 1, j], acc[i, j - 1]
            if c0 < c1 and c0 < c2:
                c, t = c0, 0
            elif c1 < c0 and c1 < c2:
                c, t = c1, 1
            else:
                c, t = c2, 2
            acc[i, j] = np.float32(cost[i - 1, j - 1]) + c
            trace[i, j] = t
    trace[0, :] = 2
    trace[:, 0] = 1
    i, j = N, M
    out = []
    while i > 0 or j > 0:
        out.append((i - 1, j - 1))
        t = trace[i, j]
        if t == 0:
            i, j = i - 1, j - 1
        elif t == 1:
            i -= 1
        else:
            j -= 1
    out.reverse()
    return [a for a, _ in out], [b for _, b in out]


def median_filter7(w: np.ndarray) -> np.ndarray:
    """width-7 median along the last axis with reflect padding (scipy.signal.medfilt-with-reflect, :227-247)."""
    F = w.shape[-1]
    idx = np.arange(F)[:, None] + np.arange(-3, 4)[None, :]
    idx = np.where(idx < 0, -idx, idx)
    idx = np.where(idx >= F, 2 * F - idx - 2, idx)
    idx = np.clip(idx, 0, F - 1)
    return np.median(w[..., idx], axis=-1).astype(np.float32)


def alignment_matrix(model: "WhisperOracle", xa: torch.Tensor, tokens: list[int], heads: list[tuple[int, int]], num_frames: int, eot: int):
    """findAlignment up to the DTW input: returns (matrix [n_tok, frames], token_probs [n_tok - 1])."""
    qks: list = []
    logits, _ = model.decode(tokens, xa, cross_qk_out=qks)
    F = num_frames // 2
    wts = torch.stack([qks[l][0, h] for l, h in heads])[:, :, :F]
    wts = torch.softmax(wts, dim=-1)
    mean = wts.mean(dim=-2, keepdim=True)
    var = wts.var(dim=-2, keepdim=True, unbiased=False)
    wts = ((wts - mean) / torch.sqrt(var + 1e-8)).numpy().astype(np.float32)
    mat = median_filter7(wts).mean(axis=0)
    lp = torch.softmax(logits[0, :, :eot], dim=-1).numpy()
    probs = np.asarray([lp[p, tokens[p + 1]] if tokens[p + 1] < eot else 0.0 for p in range(len(tokens) - 1)], np.float32)
    return mat, probs


def word_times(text_idx, time_idx, word_groups, tokens_per_second: float = 50.0):
    """jump / boundary arithmetic of findAlignment (:735-790): word_groups = token groups of the words + the eot group."""
    boundaries = [0]
    for g in word_groups[:-1]:
        boundaries.append(boundaries[-1] + len(g))
    jumps = [0] + [i for i in range(1, len(text_idx)) if text_idx[i] != text_idx[i - 1]]
    jt = [time_idx[i] / tokens_per_second for i in jumps]
    out = []
    for i in range(len(word_groups) - 1):
        s = jt[boundaries[i]] if boundaries[i] < len(jt) else (jt[-1] if jt else 0.0)
        e = jt[boundaries[i + 1]] if boundaries[i + 1] < len(jt) else (jt[-1] if jt else s)
        out.append((s, max(e, s)))
    return out
