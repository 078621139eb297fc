"""Host-side mirror of WhisperSTT.transcribe (STT/Whisper/WhisperSTT.swift:117-621): the seek-based loop over 30 s windows
with prompt conditioning, temperature fallback, no-speech skipping, timestamp-pair segment slicing, seek advance and the
post-filters -- for a BATCH of clips at once (the reference handles one clip; every clip here follows the batch-1 rules).

This is CPU integer/string logic in the reference too; all tensor work goes through the HIP layer via `decode_fn`:
    decode_fn(mels [n, 3000, n_mels] fp32, prompts list[list[int]], temperatures list[float], uniforms [n, max_tokens] | None)
        -> list[DecodingResult]
The text codec (tiktoken BPE) is not part of the hot path and has no vocabulary file offline: `tokenizer` is any object with
decode(list[int]) -> str; tests use a synthetic one.  Word-level timestamps (timestamps == .word) are §8(f) rank 3, not here.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field

import numpy as np

N_FRAMES = 3000
HOP_LENGTH = 160
SAMPLE_RATE = 16000
N_SAMPLES = 480000


@dataclass
class TranscriptionSegment:
    text: str
    start: float
    end: float
    tokens: list[int]
    avg_logprob: float
    no_speech_prob: float


@dataclass
class TranscriptionResult:
    text: str
    language: str | int
    segments: list[TranscriptionSegment]
    duration: float
    passes: int = 0          # decoder passes spent on this clip (windows x fallback attempts)


def compression_ratio(text: str) -> float:
    """computeCompressionRatio (WhisperDecoding.swift:421-447): Apple COMPRESSION_ZLIB == raw DEFLATE, level 5, no header."""
    if not text:
        return 1.0
    data = text.encode("utf-8")
    c = zlib.compressobj(5, zlib.DEFLATED, -15)
    n = len(c.compress(data) + c.flush())
    return len(data) / n if n > 0 else 1.0


def pad_or_trim_mel(mel: np.ndarray, length: int = N_FRAMES) -> np.ndarray:
    """padOrTrimMel (WhisperSTT.swift:624-635): pads with 0.0 in normalised units."""
    n = mel.shape[0]
    if n == length:
        return mel
    if n > length:
        return mel[:length]
    return np.concatenate([mel, np.zeros((length - n, mel.shape[1]), mel.dtype)])


_PUNCT = set("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~¡¿·…–—‘’“” \t")


@dataclass
class _ClipState:
    mel: np.ndarray
    content_frames: int
    seek: int = 0
    all_tokens: list[int] = field(default_factory=list)
    segments: list[TranscriptionSegment] = field(default_factory=list)
    prompt_reset_since: int = 0
    passes: int = 0


def _fallback_sequence(segment_duration: float) -> list[float]:
    return [0.0, 0.5, 1.0] if segment_duration < 2.0 else [0.0, 0.2, 0.4, 0.6, 0.8, 1.0]      # WhisperSTT.swift:192-194


def transcribe_batch(full_mels: list[np.ndarray], n_samples: list[int], decode_fn, tokenizer, special, *, language=0,
                     condition_on_previous_text: bool = True, no_speech_threshold: float | None = 0.6,
                     logprob_threshold: float | None = -1.0, compression_ratio_threshold: float | None = 2.4,
                     max_tokens: int = 448, rng: np.random.Generator | None = None, n_audio_ctx: int = 1500) -> list[TranscriptionResult]:
    """full_mels[b]: log-mel of clip b + 30 s of zeros (WhisperSTT.swift:140-145), fp32 [frames, n_mels]; n_samples[b]: audio samples.
    rng supplies the explicit uniforms of the T>0 fallback draws (the reference uses an unseeded system RNG)."""
    n_frames = 2 * n_audio_ctx
    input_stride = n_frames // n_audio_ctx
    time_precision = float(input_stride * HOP_LENGTH) / SAMPLE_RATE
    tsb, eot = special.timestamp_begin, special.eot
    clips = [_ClipState(m, n // HOP_LENGTH) for m, n in zip(full_mels, n_samples)]
    rng = rng or np.random.default_rng(0)

    while True:
        act = [i for i, c in enumerate(clips) if c.seek < c.content_frames]
        if not act:
            break
        # ---- window of every active clip (WhisperSTT.swift:171-186)
        seg_size = {i: min(n_frames, clips[i].content_frames - clips[i].seek) for i in act}
        seg_dur = {i: seg_size[i] * HOP_LENGTH / SAMPLE_RATE for i in act}
        mels = np.stack([pad_or_trim_mel(clips[i].mel[clips[i].seek:clips[i].seek + seg_size[i]], n_frames) for i in act])
        prompts = [clips[i].all_tokens[clips[i].prompt_reset_since:] if condition_on_previous_text else [] for i in act]
        # the Swift does not truncate the prompt (appendix A3); keep the tail that still leaves room to generate
        prompts = [p[-(max_tokens // 2 - 1):] for p in prompts]
        # ---- temperature fallback (WhisperSTT.swift:195-250), batched: level k re-decodes only the clips that need it
        seqs = {i: _fallback_sequence(seg_dur[i]) for i in act}
        results: dict[int, object] = {}
        temps_used: dict[int, float] = {}
        pending = list(act)
        level = 0
        while pending:
            temps = [seqs[i][min(level, len(seqs[i]) - 1)] for i in pending]
            idx = [act.index(i) for i in pending]
            uni = rng.random((len(pending), max_tokens)).astype(np.float32) if any(t > 0 for t in temps) else None
            res = decode_fn(mels[idx], [prompts[j] for j in idx], temps, uni)
            nxt = []
            for i, t, r in zip(pending, temps, res):
                clips[i].passes += 1
                results[i], temps_used[i] = r, t
                text = tokenizer.decode([x for x in r.tokens if x < eot])
                needs = False
                if compression_ratio_threshold is not None and compression_ratio(text) > compression_ratio_threshold:
                    needs = True
                if logprob_threshold is not None and r.avg_logprob < logprob_threshold:
                    needs = True
                if no_speech_threshold is not None and r.no_speech_prob > no_speech_threshold:
                    needs = False
                if needs and level + 1 < len(seqs[i]):
                    nxt.append(i)
            pending = nxt
            level += 1

        # ---- per clip: no-speech skip, segment slicing, seek advance, filters (WhisperSTT.swift:258-596)
        for i in act:
            c, r, temp = clips[i], results[i], temps_used[i]
            time_offset = c.seek * HOP_LENGTH / SAMPLE_RATE
            segment_size, segment_duration = seg_size[i], seg_dur[i]
            if no_speech_threshold is not None:
                skip = r.no_speech_prob > no_speech_threshold
                if logprob_threshold is not None and r.avg_logprob > logprob_threshold:
                    skip = False
                if skip:
                    c.seek += segment_size
                    continue
            previous_seek = c.seek
            tokens = list(r.tokens)
            is_ts = [t >= tsb for t in tokens]
            consecutive = [k + 1 for k in range(len(is_ts) - 1) if is_ts[k] and is_ts[k + 1]]
            single_ts_ending = len(is_ts) >= 2 and (not is_ts[-2]) and is_ts[-1]
            current: list[TranscriptionSegment] = []
            if consecutive:
                slices = list(consecutive)
                if single_ts_ending:
                    slices.append(len(tokens))
                last = 0
                for cur in slices:
                    sl = tokens[last:cur]
                    if len(sl) >= 2:
                        start = time_offset + (sl[0] - tsb) * time_precision
                        end = time_offset + (sl[-1] - tsb) * time_precision
                        current.append(TranscriptionSegment(tokenizer.decode([x for x in sl if x < eot]), start, end, sl, r.avg_logprob, r.no_speech_prob))
                    last = cur
                if single_ts_ending:
                    lt = tokens[-1]
                    if lt != tsb and c.seek + (lt - tsb) * input_stride < c.content_frames:
                        c.seek += (lt - tsb) * input_stride
                    else:
                        c.seek += segment_size
                else:
                    last_pos = tokens[consecutive[-1] - 1] - tsb
                    c.seek += min(last_pos * input_stride, segment_size)
            else:
                duration = segment_duration
                ts_idx = [k for k, t in enumerate(tokens) if t >= tsb]
                if ts_idx and tokens[ts_idx[-1]] != tsb:
                    duration = (tokens[ts_idx[-1]] - tsb) * time_precision
                current.append(TranscriptionSegment(tokenizer.decode([x for x in tokens if x < eot]), time_offset, time_offset + duration, tokens,
                                                    r.avg_logprob, r.no_speech_prob))
                if single_ts_ending and ts_idx and tokens[ts_idx[-1]] != tsb:
                    adv = (tokens[ts_idx[-1]] - tsb) * input_stride
                    c.seek += adv if c.seek + adv < c.content_frames else segment_size
                else:
                    c.seek += segment_size
            c.seek = max(previous_seek, c.seek)
            # a window that yields no progress would loop forever on pathological (e.g. random-init) models; the Swift's
            # max(previousSeek, seek) has the same hazard -- advance by the window so the loop is total.
            if c.seek == previous_seek:
                c.seek += segment_size
            current = [s for s in current if s.end > s.start]
            current = [s for s in current if (s.end - time_offset) <= segment_duration + 1.0]
            if temp >= 0.8 and r.avg_logprob < -2.0:
                current = []
            kept = []
            for s in current:
                t = s.text.strip(" ")
                meaningful = bool(t) and not all(ch in _PUNCT for ch in t)
                if s.no_speech_prob > 0.9:
                    continue
                if s.start != s.end and meaningful:
                    kept.append(s)
            c.segments.extend(kept)
            for s in kept:
                c.all_tokens.extend(s.tokens)
            if not condition_on_previous_text or temp > 0.5:
                c.prompt_reset_since = len(c.all_tokens)

    out = []
    for c, n in zip(clips, n_samples):
        text = tokenizer.decode([t for t in c.all_tokens if t < eot]).strip(" ")
        out.append(TranscriptionResult(text, language, c.segments, n / SAMPLE_RATE, c.passes))
    return out


class WhisperSTT:
    """transcribe(audio:...) for a batch of clips on the HIP path (log-mel, encoder, decoder all on the GPU)."""

    def __init__(self, ctx, model, tokenizer, suppress_ids, blank_ids):
        self.ctx, self.model, self.tokenizer = ctx, model, tokenizer
        self.suppress_ids, self.blank_ids = list(suppress_ids), list(blank_ids)

    def _decode_fn(self, language_index, timestamps, max_tokens):
        from . import whisper as HW
        st = self.model.special

        def fn(mels, prompts, temps, uniforms):
            self.model.encode(mels)
            sot_seq = st.sot_sequence(language_index, "transcribe")
            inits, sot_idx = [], []
            for p in prompts:
                pre = ([st.sot_prev] + list(p)) if p else []
                sot_idx.append(len(pre))
                toks = pre + sot_seq + ([] if timestamps else [st.no_timestamps])
                inits.append(toks)
            o = HW.DecodingOptions(language_index=language_index, timestamps=timestamps, suppress_ids=self.suppress_ids, blank_ids=self.blank_ids,
                                   max_tokens=max_tokens)
            return self.model.decode_ragged(o, inits, sot_idx, temps, uniforms)
        return fn

    def transcribe(self, clips, language_index=0, timestamps=True, max_tokens=448, rng=None, **kw):
        from . import audio as A
        d = self.model.dims
        clips = [np.ascontiguousarray(c, np.float32) for c in clips]
        mels = [A.whisper_log_mel_spectrogram(self.ctx, c, d.n_mels, padding=N_SAMPLES) for c in clips]
        return transcribe_batch(mels, [c.shape[0] for c in clips], self._decode_fn(language_index, timestamps, max_tokens), self.tokenizer,
                                self.model.special, language=language_index, max_tokens=max_tokens, rng=rng, n_audio_ctx=d.n_audio_ctx, **kw)
