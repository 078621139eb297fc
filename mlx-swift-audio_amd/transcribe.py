"""Host-side mirror of WhisperSTT.transcribe (STT/Whisper/WhisperSTT.swift:117-621): the seek-based loop over 30 s windows
with prompt conditioning, temperature fallback, no-speech skipping, timestamp-pair segment slicing, seek advance and the
post-filters -- for a BATCH of clips at once (the reference handles one clip; every clip here follows the batch-1 rules).

This is CPU integer/string logic in the reference too; all tensor work goes through the HIP layer via `decode_fn`:
    decode_fn(mels [n, 3000, n_mels] fp32, prompts list[list[int]], temperatures list[float], uniforms [n, max_tokens] | None)
        -> list[DecodingResult]
The text codec (tiktoken BPE) is not part of the hot path and has no vocabulary file offline: `tokenizer` is any object with
decode(list[int]) -> str; tests use a synthetic one.

timestamps == .word (WhisperSTT.swift:440-540): after a window's segments are sliced, ONE alignment call per window
(`align_fn`, batched over the clips that need it: encoder pass + teacher-forced decoder pass + DTW on the GPU, timing.find_alignment)
gives the words; add_word_timestamps deals them to the segments; seek then follows the last word's end, and with
`hallucination_silence_threshold` the anomaly / surrounding-silence rules skip or drop hallucinated segments.
language = None (:155-161, :662-695): `detect_fn` runs the [sot] probe on every clip's first window and each clip decodes with its
own language token.
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
    words: list | None = None        # timing.Word list when timestamps == .word


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
    last_speech: float = 0.0         # lastSpeechTimestamp (WhisperSTT.swift:169)
    language: int = 0
    language_prob: float | None = None


def _fallback_sequence(segment_duration: float) -> list[float]:
    return [0.0, 0.5, 1.0] if segment_duration < 2.0 else [0.0, 0.2, 0.4, 0.6, 0.8, 1.0]      # WhisperSTT.swift:192-194


FRAMES_PER_SECOND = 100
_f32 = np.float32


def get_last_word_end(segments):
    """getLastWordEnd (WhisperTiming.swift:1066-1074)."""
    for seg in reversed(segments):
        if seg.words:
            return float(_f32(seg.words[-1].end))
    return float(_f32(segments[-1].end)) if segments else None


def _call_decode(decode_fn, mels, prompts, temps, uni, langs, per_clip_lang):
    return decode_fn(mels, prompts, temps, uni, langs) if per_clip_lang else decode_fn(mels, prompts, temps, uni)


def transcribe_batch(full_mels: list[np.ndarray], n_samples: list[int], decode_fn, tokenizer, special, *, language=0,
                     condition_on_previous_text: bool = True, no_speech_threshold: float | None = 0.6,
                     logprob_threshold: float | None = -1.0, compression_ratio_threshold: float | None = 2.4,
                     max_tokens: int = 448, rng: np.random.Generator | None = None, n_audio_ctx: int = 1500,
                     word_timestamps: bool = False, align_fn=None, hallucination_silence_threshold: float | None = None,
                     detect_fn=None) -> list[TranscriptionResult]:
    """full_mels[b]: log-mel of clip b + 30 s of zeros (WhisperSTT.swift:140-145), fp32 [frames, n_mels]; n_samples[b]: audio samples.
    rng supplies the explicit uniforms of the T>0 fallback draws (the reference uses an unseeded system RNG).
    language: an index for every clip, a list of per-clip indices, or None = detect (detect_fn(mels [n, 3000, n_mels]) -> [(index, prob)]);
      with per-clip languages decode_fn / align_fn receive the list as a last argument.
    word_timestamps: timestamps == .word; align_fn(mels [n, 3000, n_mels], text_tokens, num_frames[, languages]) -> list of
      timing.WordTiming lists (findAlignment of every listed clip's window)."""
    from . import timing as T
    n_frames = 2 * n_audio_ctx
    input_stride = n_frames // n_audio_ctx
    time_precision = float(input_stride * HOP_LENGTH) / SAMPLE_RATE
    tsb, eot = special.timestamp_begin, special.eot
    clips = [_ClipState(m, n // HOP_LENGTH) for m, n in zip(full_mels, n_samples)]
    rng = rng or np.random.default_rng(0)
    hst = hallucination_silence_threshold
    if word_timestamps and align_fn is None:
        raise ValueError("word_timestamps needs align_fn")
    # ---- language (WhisperSTT.swift:155-161): detect on the first window of every clip when not given
    per_clip_lang = language is None or isinstance(language, (list, tuple))
    if language is None:
        if detect_fn is None:
            raise ValueError("language=None needs detect_fn")
        first = np.stack([pad_or_trim_mel(c.mel[:n_frames], n_frames) for c in clips])
        for c, (li, lp) in zip(clips, detect_fn(first)):
            c.language, c.language_prob = int(li), float(lp)
    elif per_clip_lang:
        for c, li in zip(clips, language):
            c.language = int(li)
    else:
        for c in clips:
            c.language = language

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
            res = _call_decode(decode_fn, mels[idx], [prompts[j] for j in idx], temps, uni, [clips[i].language for i in pending], per_clip_lang)
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

        # ---- phase 1, per clip: no-speech skip, segment slicing, seek advance, window filters (WhisperSTT.swift:258-438)
        staged: dict[int, dict] = {}
        for i in act:
            c, r, temp = clips[i], results[i], temps_used[i]
            time_offset = c.seek * HOP_LENGTH / SAMPLE_RATE
            segment_size, segment_duration = seg_size[i], seg_dur[i]
            window_end_time = (c.seek + n_frames) * HOP_LENGTH / SAMPLE_RATE
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
            current = [s for s in current if s.end > s.start]
            current = [s for s in current if (s.end - time_offset) <= segment_duration + 1.0]
            if temp >= 0.8 and r.avg_logprob < -2.0:
                current = []
            staged[i] = dict(current=current, previous_seek=previous_seek, time_offset=time_offset, segment_size=segment_size,
                             segment_duration=segment_duration, window_end_time=window_end_time, single_ts_ending=single_ts_ending, temp=temp)

        # ---- phase 2: ONE alignment call per window for the clips that have text (addWordTimestamps' single findAlignment, :441-452)
        alignments: dict[int, list] = {}
        if word_timestamps:
            want = [i for i in staged if any(t < eot for s in staged[i]["current"] for t in s.tokens)]
            if want:
                idx = [act.index(i) for i in want]
                text_tokens = [[t for s in staged[i]["current"] for t in s.tokens if t < eot] for i in want]
                args = (mels[idx], text_tokens, [seg_size[i] for i in want])
                got = align_fn(*args, [clips[i].language for i in want]) if per_clip_lang else align_fn(*args)
                alignments = dict(zip(want, got))

        # ---- phase 3, per clip: word timestamps, hallucination rules, final filters (WhisperSTT.swift:440-596)
        for i, stg in staged.items():
            c = clips[i]
            current, previous_seek, time_offset = stg["current"], stg["previous_seek"], stg["time_offset"]
            segment_size, segment_duration, window_end_time = stg["segment_size"], stg["segment_duration"], stg["window_end_time"]
            single_ts_ending, temp = stg["single_ts_ending"], stg["temp"]
            skip_window = False
            if word_timestamps:
                c.last_speech = T.add_word_timestamps(current, alignments.get(i, []), eot, time_offset, c.last_speech)
                f_off = _f32(time_offset)
                if not single_ts_ending:
                    lwe = get_last_word_end(current)
                    if lwe is not None and _f32(lwe) > f_off:
                        c.seek = int(_f32(lwe) * _f32(FRAMES_PER_SECOND))
                if hst is not None:
                    thr = _f32(hst)
                    if not single_ts_ending:
                        lwe = get_last_word_end(current)
                        if lwe is not None and _f32(lwe) > f_off:
                            remaining = _f32(window_end_time) - _f32(lwe)
                            c.seek = int(_f32(lwe) * _f32(FRAMES_PER_SECOND)) if remaining > thr else previous_seek + segment_size
                    first = next((s for s in current if s.words), None)
                    if first is not None and T.is_segment_anomaly(first.words):
                        gap = _f32(first.start) - f_off
                        if gap > thr:                         # leading-silence hallucination: skip ahead, keep nothing of this window
                            c.seek = previous_seek + int(gap * _f32(FRAMES_PER_SECOND))
                            skip_window = True
                    if not skip_window:
                        hal_last_end = _f32(c.last_speech)
                        for si, seg in enumerate(current):
                            if not seg.words:
                                continue
                            if T.is_segment_anomaly(seg.words):
                                nxt_seg = next((s for s in current[si + 1:] if s.words), None)
                                hal_next_start = _f32(nxt_seg.words[0].start) if nxt_seg is not None else f_off + _f32(segment_duration)
                                s0, s1 = _f32(seg.start), _f32(seg.end)
                                silence_before = (s0 - hal_last_end > thr) or (s0 < thr) or (s0 - f_off < _f32(2.0))
                                silence_after = (hal_next_start - s1 > thr) or T.is_segment_anomaly(nxt_seg.words if nxt_seg is not None else None) or \
                                    (_f32(window_end_time) - s1 < _f32(2.0))
                                if silence_before and silence_after:
                                    c.seek = int(max(f_off + _f32(1), s0) * _f32(FRAMES_PER_SECOND))
                                    if _f32(c.content_frames * HOP_LENGTH / SAMPLE_RATE) - s1 < thr:
                                        c.seek = c.content_frames
                                    del current[si:]
                                    break
                            hal_last_end = _f32(seg.end)
                if not skip_window:
                    lwe = get_last_word_end(current)
                    if lwe is not None:
                        c.last_speech = lwe
            # a window that yields no progress would loop forever on pathological (e.g. random-init) models; the Swift's
            # max(previousSeek, seek) has the same hazard -- advance by the window so the loop is total.
            if c.seek <= previous_seek:
                c.seek = previous_seek + segment_size
            if skip_window:
                continue
            kept = []
            for s in current:
                t = s.text.strip(" ")
                meaningful = bool(t) and not all(ch in _PUNCT for ch in t)
                if s.no_speech_prob > 0.9:
                    continue
                if word_timestamps:
                    if meaningful and not s.words and len(t) > 10:       # text without any aligned word: a likely hallucination
                        continue
                    if s.words and T.is_segment_anomaly(s.words):
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
        out.append(TranscriptionResult(text, c.language, c.segments, n / SAMPLE_RATE, c.passes))
    return out


class WhisperSTT:
    """transcribe(audio:...) for a batch of clips on the HIP path (log-mel, encoder, decoder, alignment all on the GPU)."""

    def __init__(self, ctx, model, tokenizer, suppress_ids, blank_ids, alignment_heads=None, split_to_word_tokens=None):
        self.ctx, self.model, self.tokenizer = ctx, model, tokenizer
        self.suppress_ids, self.blank_ids = list(suppress_ids), list(blank_ids)
        self.alignment_heads = alignment_heads            # [(layer, head)]: the checkpoint's alignment_heads (WhisperModel.swift:95-99)
        self.split_to_word_tokens = split_to_word_tokens  # tokens + [eot] -> (words, token groups) (WhisperTokenizer.swift:546-670)

    def _decode_fn(self, language_index, timestamps, max_tokens):
        from . import whisper as HW
        st = self.model.special

        def fn(mels, prompts, temps, uniforms, langs=None):
            self.model.encode(mels)
            inits, sot_idx = [], []
            for b, p in enumerate(prompts):
                sot_seq = st.sot_sequence(language_index if langs is None else langs[b], "transcribe")
                pre = ([st.sot_prev] + list(p)) if p else []
                sot_idx.append(len(pre))
                toks = pre + sot_seq + ([] if timestamps else [st.no_timestamps])
                inits.append(toks)
            o = HW.DecodingOptions(language_index=language_index, timestamps=timestamps, suppress_ids=self.suppress_ids, blank_ids=self.blank_ids,
                                   max_tokens=max_tokens)
            return self.model.decode_ragged(o, inits, sot_idx, temps, uniforms)
        return fn

    def _align_fn(self, language_index):
        from . import timing as T
        st = self.model.special

        def fn(mels, text_tokens, num_frames, langs=None):
            # findAlignment runs its own encoder pass on the window (model.forwardWithCrossQK, WhisperTiming.swift:590-598)
            self.model.encode(mels)
            if langs is None or len(set(langs)) <= 1:
                li = language_index if langs is None else langs[0]
                return T.find_alignment(self.model, text_tokens, num_frames, st.sot_sequence(li, "transcribe"), st, self.alignment_heads,
                                        self.split_to_word_tokens)
            # clips with different language tokens: the sot sequence is part of the teacher-forced prefix, one call per language
            out = [None] * len(text_tokens)
            for li in sorted(set(langs)):
                sel = [b for b, l in enumerate(langs) if l == li]
                self.model.encode(mels[sel])
                for b, wt in zip(sel, T.find_alignment(self.model, [text_tokens[b] for b in sel], [num_frames[b] for b in sel],
                                                       st.sot_sequence(li, "transcribe"), st, self.alignment_heads, self.split_to_word_tokens)):
                    out[b] = wt
            return out
        return fn

    def _detect_fn(self):
        def fn(mels):
            self.model.encode(mels)
            return self.model.detect_language()
        return fn

    def transcribe(self, clips, language_index=0, timestamps=True, max_tokens=448, rng=None, word_timestamps=False, **kw):
        """language_index None = detect per clip (WhisperSTT.swift:155-161); word_timestamps = timestamps == .word (needs the
        checkpoint's alignment_heads and a word splitter, given to the constructor)."""
        from . import audio as A
        d = self.model.dims
        clips = [np.ascontiguousarray(c, np.float32) for c in clips]
        mels = [A.whisper_log_mel_spectrogram(self.ctx, c, d.n_mels, padding=N_SAMPLES) for c in clips]
        if word_timestamps and (self.alignment_heads is None or self.split_to_word_tokens is None):
            raise ValueError("word_timestamps needs alignment_heads and split_to_word_tokens")
        return transcribe_batch(mels, [c.shape[0] for c in clips], self._decode_fn(language_index, timestamps, max_tokens), self.tokenizer,
                                self.model.special, language=language_index, max_tokens=max_tokens, rng=rng, n_audio_ctx=d.n_audio_ctx,
                                word_timestamps=word_timestamps, align_fn=self._align_fn(language_index) if word_timestamps else None,
                                detect_fn=self._detect_fn() if language_index is None else None, **kw)
