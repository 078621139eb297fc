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


# ---- the back half of the word-timestamp path: text / float host logic of WhisperTiming.swift:311-556, 847-1060 ---------------------------
# (no tensors here: everything below consumes the WordTiming lists find_alignment() returns; Float arithmetic is kept in numpy float32
# where the Swift uses Float, so thresholds compare the same way)
import unicodedata  # noqa: E402

# The Swift literals are RAW strings (#"..."#), so their leading \" is a backslash followed by a quote, and the quote after the
# apostrophe is the ASCII one (upstream Python has curly quotes there): restated byte for byte.
DEFAULT_PREPEND_PUNCTUATIONS = '\\"\'"\u00bf([{-'                                            # WhisperTiming.swift:314
DEFAULT_APPEND_PUNCTUATIONS = '\\"\'.\u3002,\uff0c!\uff01?\uff1f:\uff1a")]}\u3001'      # :317
SENTENCE_END_MARKS = {".", "。", "!", "！", "?", "？"}      # :377
PYTHON_PUNCTUATION = "!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~"    # :1006

_f32 = np.float32


def _is_spaceless_script_char(ch: str) -> bool:
    """WhisperTokenizer.swift:575-586."""
    v = ord(ch)
    return (0x4E00 <= v <= 0x9FFF or 0x3400 <= v <= 0x4DBF or 0x20000 <= v <= 0x2A6DF or 0x3040 <= v <= 0x309F or 0x30A0 <= v <= 0x30FF
            or 0xAC00 <= v <= 0xD7AF or 0x0E00 <= v <= 0x0E7F or 0x0E80 <= v <= 0x0EFF or 0x1000 <= v <= 0x109F)


def make_split_to_word_tokens(decode, eot: int):
    """splitToWordTokens (WhisperTokenizer.swift:546-670) over a caller-supplied `decode(tokens) -> str` (the BPE text codec itself is
    outside the hot path).  Returns the callable find_alignment() expects: tokens -> (words, token groups)."""

    def split_character_level(tokens):
        words, groups = [], []
        for t in tokens:
            if t >= eot:
                words.append("")
                groups.append([t])
                continue
            d = decode([t])
            if d:
                words.append(d)
                groups.append([t])
        return words, groups

    def split_by_whitespace(tokens):
        words, groups = [], []
        cur_w, cur_t = "", []
        hit_eot = False
        for t in tokens:
            if t >= eot:
                if cur_w or cur_t:
                    words.append(cur_w)
                    groups.append(cur_t)
                words.append("")
                groups.append([t])
                hit_eot = True
                cur_w, cur_t = "", []
                break
            d = decode([t])
            if d.startswith(" ") and (cur_w or cur_t):
                words.append(cur_w)
                groups.append(cur_t)
                cur_w, cur_t = d, [t]
            else:
                cur_w += d
                cur_t = cur_t + [t]
        if not hit_eot and (cur_w or cur_t):
            words.append(cur_w)
            groups.append(cur_t)
        return words, groups

    def split(tokens):
        text_tokens = [t for t in tokens if t < eot]
        if not text_tokens:
            return ([""], [[eot]]) if tokens and tokens[-1] == eot else ([], [])
        decoded = decode(text_tokens)
        n_spaceless = sum(1 for ch in decoded if _is_spaceless_script_char(ch))
        if n_spaceless > len(decoded) // 2:
            return split_character_level(tokens)
        return split_by_whitespace(tokens)

    return split


def merge_punctuations(alignment: list[WordTiming], prepended: str = DEFAULT_PREPEND_PUNCTUATIONS, appended: str = DEFAULT_APPEND_PUNCTUATIONS) -> None:
    """mergePunctuations (WhisperTiming.swift:328-371), in place."""
    if len(alignment) <= 1:
        return
    i, j = len(alignment) - 2, len(alignment) - 1
    while i >= 0:
        prev = alignment[i].word
        # Swift: `prepended.contains(previousWord.trimmingCharacters(in: .whitespaces))` -- Foundation's StringProtocol.contains(String):
        # a substring test that is FALSE for the empty string (range(of: "") is nil), unlike Python's `"" in s`
        trimmed = prev.strip(" \t\u00a0")
        if prev.startswith(" ") and trimmed != "" and trimmed in prepended:
            alignment[j].word = prev + alignment[j].word
            alignment[j].tokens = alignment[i].tokens + alignment[j].tokens
            alignment[j].start = alignment[i].start
            alignment[i].word = ""
            alignment[i].tokens = []
        else:
            j = i
        i -= 1
    i, j = 0, 1
    while j < len(alignment):
        prev, foll = alignment[i].word, alignment[j].word
        if not prev.endswith(" ") and foll != "" and foll in appended:      # (an emptied slot is never "contained": see above)
            alignment[i].word = prev + foll
            alignment[i].tokens = alignment[i].tokens + alignment[j].tokens
            alignment[i].end = alignment[j].end
            alignment[j].word = ""
            alignment[j].tokens = []
        else:
            i = j
        j += 1
    alignment[:] = [a for a in alignment if not (a.word == "" and not a.tokens)]


def calculate_duration_thresholds(alignment: list[WordTiming]) -> tuple[float, float]:
    """calculateDurationThresholds (:386-408): (min(0.7, median of the non-zero durations), twice that)."""
    d = sorted(_f32(a.end) - _f32(a.start) for a in alignment if _f32(a.end) - _f32(a.start) > 0)
    if not d:
        return 0.0, 0.0
    n = len(d)
    med = (d[n // 2 - 1] + d[n // 2]) / _f32(2) if n % 2 == 0 else d[n // 2]
    capped = min(_f32(0.7), _f32(med))
    return float(capped), float(capped * _f32(2))


def clip_at_sentence_boundaries(alignment: list[WordTiming], max_duration: float) -> None:
    """clipAtSentenceBoundaries (:418-441), in place."""
    if len(alignment) <= 1 or max_duration <= 0:
        return
    for i in range(1, len(alignment)):
        if _f32(alignment[i].end) - _f32(alignment[i].start) <= _f32(max_duration):
            continue
        if alignment[i].word in SENTENCE_END_MARKS:
            alignment[i].end = float(_f32(alignment[i].start) + _f32(max_duration))
        elif alignment[i - 1].word in SENTENCE_END_MARKS:
            alignment[i].start = float(_f32(alignment[i].end) - _f32(max_duration))


def clip_at_segment_boundaries(words: list[WordTiming], last_speech_timestamp: float, median_duration: float, max_duration: float) -> None:
    """clipAtSegmentBoundaries (:453-485), in place."""
    if not words or max_duration <= 0:
        return
    first = words[0]
    if _f32(first.end) - _f32(last_speech_timestamp) > _f32(median_duration) * _f32(4):
        needs = (_f32(first.end) - _f32(first.start) > _f32(max_duration)) or (len(words) > 1 and _f32(words[1].end) - _f32(first.start) > _f32(max_duration) * _f32(2))
        if needs:
            if len(words) > 1 and _f32(words[1].end) - _f32(words[1].start) > _f32(max_duration):
                boundary = max(_f32(words[1].end) / _f32(2), _f32(words[1].end) - _f32(max_duration))
                words[0].end = float(boundary)
                words[1].start = float(boundary)
            words[0].start = float(max(_f32(0), _f32(words[0].end) - _f32(max_duration)))


@dataclass
class Word:
    word: str
    start: float
    end: float
    probability: float


def add_word_timestamps(segments: list, alignment: list[WordTiming], eot: int, time_offset: float, last_speech_timestamp: float) -> float:
    """addWordTimestamps (:847-1002) after its single findAlignment call: `alignment` is find_alignment()'s list for the concatenated
    text tokens (token < eot) of `segments` (objects with .tokens, .start, .end and a writable .words).  Clips long words at sentence
    ends, merges punctuation, deals the words back to their segments, clips after pauses and adjusts the segment bounds.
    Returns the updated last_speech_timestamp."""
    if not segments:
        return last_speech_timestamp
    per_segment = [[t for t in s.tokens if t < eot] for s in segments]
    if not any(per_segment) or not alignment:
        return last_speech_timestamp
    median_d, max_d = calculate_duration_thresholds(alignment)
    if max_d > 0:
        clip_at_sentence_boundaries(alignment, max_d)
    merge_punctuations(alignment)
    wi = 0
    last = last_speech_timestamp
    for si, toks in enumerate(per_segment):
        saved = 0
        words: list[WordTiming] = []
        while wi < len(alignment) and saved < len(toks):
            t = alignment[wi]
            if t.word != "":
                words.append(WordTiming(t.word, [], float(_f32(time_offset) + _f32(t.start)), float(_f32(time_offset) + _f32(t.end)), t.probability))
            saved += len(t.tokens)
            wi += 1
        if not words:
            continue
        if max_d > 0:
            clip_at_segment_boundaries(words, last, median_d, max_d)
        seg = segments[si]
        s0, s1 = _f32(seg.start), _f32(seg.end)
        first, lastw = words[0], words[-1]
        if s0 < _f32(first.end) and s0 - _f32(0.5) > _f32(first.start):
            a_start = max(_f32(0), min(_f32(first.end) - _f32(median_d), s0))
            first.start = float(a_start)
        else:
            a_start = _f32(first.start)
        if s1 > _f32(lastw.start) and s1 + _f32(0.5) < _f32(lastw.end):
            a_end = max(_f32(lastw.start) + _f32(median_d), s1)
            lastw.end = float(a_end)
        else:
            a_end = _f32(lastw.end)
        last = float(a_end)
        seg.start, seg.end = float(a_start), float(a_end)
        seg.words = [Word(w.word, w.start, w.end, w.probability) for w in words]
    return last


def word_anomaly_score(word) -> float:
    """wordAnomalyScore (:1016-1037)."""
    d = _f32(word.end) - _f32(word.start)
    score = _f32(0)
    if word.probability < 0.15:
        score += _f32(1)
    if d < _f32(0.133):
        score += (_f32(0.133) - d) * _f32(15)
    if d > _f32(2.0):
        score += d - _f32(2.0)
    return float(score)


def is_segment_anomaly(words) -> bool:
    """isSegmentAnomaly (:1040-1060): the first 8 words that are not a substring of Python's string.punctuation."""
    if not words:
        return False
    kept = [w for w in words if w.word not in PYTHON_PUNCTUATION][:8]
    if not kept:
        return False
    score = _f32(0)
    for w in kept:
        score += _f32(word_anomaly_score(w))
    return bool(score >= 3 or score + _f32(0.01) >= _f32(len(kept)))
