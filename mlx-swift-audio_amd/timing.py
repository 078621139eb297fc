"""Host-side mirror of the word-timestamp path (STT/Whisper/WhisperTiming.swift): findAlignment (:558-820) with its tensor half on the
GPU (mia_whisper_align: teacher-forced pass, alignment-head QK, softmax / standardise / median-7 / head mean, DTW) and the word /
jump-time arithmetic here.  Splitting tokens into words needs the tokenizer's text (`split_to_word_tokens`, a caller-supplied
callable: tokens + [eot] -> (words, token groups), WhisperTokenizer.swift) -- the text codec itself is outside the hot path."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib

TOKENS_PER_SECOND = 50.0


@dataclass
class WordTiming:
    word: str
    tokens: list[int]
    start: float
    end: float
    probability: float


def _declare(lib):
    if getattr(lib, "_align_declared", False):
        return
    vp, i32 = C.c_void_p, C.c_int
    lib.mia_whisper_align.restype = i32
    lib.mia_whisper_align.argtypes = [vp, vp, i32, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp, i32, vp]
    lib._align_declared = True


def align(model, token_seqs: list[list[int]], heads, num_frames: list[int], row_start: int, eot: int, want_matrix: bool = False):
    """Raw call: one token sequence per clip of the last encode().  Returns (token_probs [B, stride], paths [(text_idx, time_idx)], matrix | None)."""
    lib = model.ctx.lib
    _declare(lib)
    B = len(token_seqs)
    stride = max(len(t) for t in token_seqs)
    toks = np.zeros((B, stride), np.int32)
    n = np.zeros(B, np.int32)
    for b, t in enumerate(token_seqs):
        toks[b, :len(t)] = t
        n[b] = len(t)
    h = np.ascontiguousarray(heads, np.int32).reshape(-1, 2)
    nf = np.ascontiguousarray(num_frames, np.int32)
    cap = stride + model.dims.n_audio_ctx + 2
    probs = np.zeros((B, stride), np.float32)
    ti = np.zeros((B, cap), np.int32)
    tj = np.zeros((B, cap), np.int32)
    pl = np.zeros(B, np.int32)
    mat = np.zeros((B, stride, model.dims.n_audio_ctx), np.float32) if want_matrix else None
    model.ctx.check(lib.mia_whisper_align(model.h, toks.ctypes.data, stride, n.ctypes.data, h.ctypes.data, h.shape[0], nf.ctypes.data, row_start, eot,
                                          probs.ctypes.data, ti.ctypes.data, tj.ctypes.data, pl.ctypes.data, cap, mat.ctypes.data if want_matrix else None))
    paths = [(ti[b, :pl[b]].tolist(), tj[b, :pl[b]].tolist()) for b in range(B)]
    return probs, paths, mat


def find_alignment(model, text_tokens: list[list[int]], num_frames: list[int], sot_sequence: list[int], special, heads, split_to_word_tokens) -> list[list[WordTiming]]:
    """findAlignment for every clip of the last encode() (WhisperTiming.swift:558-820)."""
    seqs, active = [], []
    for b, tt in enumerate(text_tokens):
        seqs.append(list(sot_sequence) + [special.no_timestamps] + list(tt) + [special.eot])
        active.append(len(tt) > 0 and num_frames[b] // 2 >= 2)
    no_ts_index, text_start = len(sot_sequence), len(sot_sequence) + 1
    probs, paths, _ = align(model, seqs, heads, [max(f, 4) for f in num_frames], no_ts_index, special.eot)
    out: list[list[WordTiming]] = []
    for b, tt in enumerate(text_tokens):
        if not active[b]:
            out.append([])
            continue
        text_idx, time_idx = paths[b]
        words, groups = split_to_word_tokens(list(tt) + [special.eot])
        if len(groups) <= 1:
            out.append([])
            continue
        boundaries = [0]
        for g in groups[:-1]:
            boundaries.append(boundaries[-1] + len(g))
        jumps = [0] + [i for i in range(1, len(text_idx)) if text_idx[i] != text_idx[i - 1]]
        jt = [time_idx[i] / TOKENS_PER_SECOND for i in jumps]
        tprob = probs[b, text_start - 1:text_start - 1 + len(tt)]
        res = []
        for i in range(len(words) - 1):
            s = jt[boundaries[i]] if boundaries[i] < len(jt) else (jt[-1] if jt else 0.0)
            e = jt[boundaries[i + 1]] if boundaries[i + 1] < len(jt) else (jt[-1] if jt else s)
            p0, p1 = boundaries[i], min(boundaries[i + 1], len(tprob))
            res.append(WordTiming(words[i], list(groups[i]), float(s), float(max(e, s)), float(tprob[p0:p1].mean()) if p0 < p1 else 0.0))
        out.append(res)
    return out
