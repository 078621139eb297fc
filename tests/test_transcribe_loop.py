"""CPU tests of the host transcribe loop (mlx-swift-audio_amd/transcribe.py) against the rules stated in
WhisperSTT.swift:171-600, with a scripted decoder standing in for the device path."""
import numpy as np

from mlx_swift_audio_amd import transcribe as T
from mlx_swift_audio_amd.whisper import DecodingResult, SpecialTokens


class FakeTok:
    def decode(self, toks):
        return "".join(" w%d" % t for t in toks)


ST = SpecialTokens.for_vocab(51864)
TSB = ST.timestamp_begin


def _run(script, n_samples=480000 * 2, **kw):
    """script: list of DecodingResult returned for successive decode calls (batch of 1)."""
    calls = []

    def decode_fn(mels, prompts, temps, uniforms):
        assert mels.shape[1:] == (3000, 80)
        calls.append((list(prompts[0]), temps[0], uniforms is not None))
        return [script[min(len(calls) - 1, len(script) - 1)]]

    mel = np.zeros((n_samples // 160 + 3000, 80), np.float32)
    res = T.transcribe_batch([mel], [n_samples], decode_fn, FakeTok(), ST, **kw)[0]
    return res, calls


def test_consecutive_timestamp_pairs_slice_segments_and_advance_seek():
    # <0.00> a b <2.00><2.00> c <5.00><5.00>  -> two segments, seek advances by the LAST consecutive pair's first stamp (5.00 s = 500 frames)
    toks = [TSB, 10, 11, TSB + 100, TSB + 100, 12, TSB + 250, TSB + 250]
    res, calls = _run([DecodingResult(toks, -0.3, 0.01)], n_samples=160 * 1000)
    segs = res.segments
    assert [(round(s.start, 2), round(s.end, 2)) for s in segs[:2]] == [(0.0, 2.0), (2.0, 5.0)]
    assert segs[0].tokens == toks[:4] and segs[1].tokens == toks[4:7]
    # second window starts at 5.00 s and is conditioned on the kept tokens of the first
    assert calls[1][0] == toks[:7]
    assert abs(segs[2].start - 5.0) < 1e-6


def test_no_speech_skip_and_logprob_override():
    res, calls = _run([DecodingResult([TSB, 10, TSB + 50], -1.5, 0.9)], n_samples=480000)
    assert res.segments == [] and len(calls) == 1                # skipped, no fallback (no-speech accepted)
    res, _ = _run([DecodingResult([TSB, 10, TSB + 50], -0.5, 0.7)], n_samples=16000)
    assert len(res.segments) == 1                               # avg_logprob above the threshold keeps it


def test_temperature_fallback_sequence_and_prompt_reset():
    bad = DecodingResult([TSB, 10, TSB + 50], -1.6, 0.1)       # low confidence -> retry
    good = DecodingResult([TSB, 10, 11, TSB + 1500], -0.2, 0.1)
    res, calls = _run([bad, bad, bad, good], n_samples=480000)
    assert [round(c[1], 1) for c in calls][:4] == [0.0, 0.2, 0.4, 0.6]
    assert calls[0][2] is False and calls[1][2] is True        # uniforms only when sampling
    assert len(res.segments) == 1 and res.passes == 4 and len(calls) == 4
    # short window (< 2 s): 3-step sequence
    _, calls = _run([bad], n_samples=16000)
    assert [round(c[1], 1) for c in calls] == [0.0, 0.5, 1.0]


def test_single_timestamp_ending_and_hallucinated_timestamp_filter():
    toks = [TSB, 10, 11, TSB + 200]                            # text then one closing timestamp at 4.00 s
    res, calls = _run([DecodingResult(toks, -0.2, 0.0)], n_samples=160 * 3000 + 160 * 500)
    assert abs(res.segments[0].end - 4.0) < 1e-6
    assert abs(res.segments[1].start - 4.0) < 1e-6             # seek advanced to the timestamp, not the full window
    # a segment whose end exceeds the window by > 1 s is dropped (:417-426)
    res, _ = _run([DecodingResult([TSB, 10, TSB + 1000], -0.2, 0.0)], n_samples=16000 * 3)
    assert res.segments == []


def test_compression_ratio_matches_raw_deflate():
    assert T.compression_ratio("") == 1.0
    assert T.compression_ratio("ab" * 200) > 2.4 > T.compression_ratio("the quick brown fox jumps over the lazy dog")


# ---- timestamps == .word and language detection (WhisperSTT.swift:155-161, 440-590) -------------------------------------------------
from mlx_swift_audio_amd.timing import WordTiming  # noqa: E402


def _run_words(script, word_script, n_samples=480000 * 2, **kw):
    """word_script: per decode window, a list of (start, end, probability) relative to the window, one per text token."""
    calls, aligns = [], []

    def decode_fn(mels, prompts, temps, uniforms):
        calls.append((list(prompts[0]), temps[0]))
        return [script[min(len(calls) - 1, len(script) - 1)]]

    def align_fn(mels, text_tokens, num_frames):
        assert mels.shape[1:] == (3000, 80) and len(text_tokens) == 1
        spec = word_script[min(len(aligns), len(word_script) - 1)]
        aligns.append((list(text_tokens[0]), num_frames[0]))
        assert len(spec) >= len(text_tokens[0])          # later (shorter) windows may have lost a segment to the window filters
        return [[WordTiming(" w%d" % t, [t], s, e, p) for t, (s, e, p) in zip(text_tokens[0], spec)]]

    mel = np.zeros((n_samples // 160 + 3000, 80), np.float32)
    res = T.transcribe_batch([mel], [n_samples], decode_fn, FakeTok(), ST, word_timestamps=True, align_fn=align_fn, **kw)[0]
    return res, calls, aligns


def test_word_timestamps_attach_words_and_seek_follows_last_word():
    # <0.00> a b <2.00><2.00> c <5.00><5.00>: not single-timestamp-ending -> seek = last word end (4.2 s = 420 frames), not the 5.00 stamp
    toks = [TSB, 10, 11, TSB + 100, TSB + 100, 12, TSB + 250, TSB + 250]
    words = [(0.2, 0.8, 0.9), (0.9, 1.7, 0.9), (2.4, 4.2, 0.9)]
    res, calls, aligns = _run_words([DecodingResult(toks, -0.3, 0.01)], [words], n_samples=160 * 1000)
    assert aligns[0] == ([10, 11, 12], 1000)                                   # ONE alignment call for the window's text tokens
    s0, s1 = res.segments[0], res.segments[1]
    assert [w.word for w in s0.words] == [" w10", " w11"] and [w.word for w in s1.words] == [" w12"]
    assert abs(s0.words[0].start - 0.2) < 1e-6 and abs(s1.words[0].end - 4.2) < 1e-6
    assert len(aligns) >= 2 and calls[1][0] == toks[:7]                        # next window decoded ...
    # ... and it starts at the last word's end (Int(4.2f * 100) = 419 frames in Float arithmetic): its words are offset by 4.19 s
    assert abs(res.segments[2].words[0].start - (4.19 + 0.2)) < 1e-4


def test_hallucination_threshold_resets_seek_when_little_silence_remains():
    # 10 s of audio, one full window; last word ends at 29.5 s of the 30 s window -> remaining 0.5 s <= threshold 2 -> seek = previous + segment_size
    toks = [TSB, 10, 11, TSB + 100, TSB + 100, 12, TSB + 250, TSB + 250]
    words = [(0.2, 0.8, 0.9), (0.9, 1.7, 0.9), (2.4, 4.2, 0.9)]
    res, calls, aligns = _run_words([DecodingResult(toks, -0.3, 0.01)], [words], n_samples=160 * 1000, hallucination_silence_threshold=2.0)
    # window_end_time = 30 s, last word 4.2 s -> remaining 25.8 > 2: seek stays at the last word end (second window decoded)
    assert len(calls) >= 2
    words_late = [(0.2, 0.8, 0.9), (0.9, 1.7, 0.9), (2.4, 29.5, 0.9)]
    toks_late = [TSB, 10, 11, TSB + 100, TSB + 100, 12, TSB + 1490, TSB + 1490]
    res2, calls2, _ = _run_words([DecodingResult(toks_late, -0.3, 0.01)], [words_late], n_samples=480000, hallucination_silence_threshold=2.0)
    assert len(calls2) == 1                                                    # seek jumped to the end of the (only) window


def test_leading_silence_hallucination_skips_the_window():
    # the first segment's words are anomalous (very short, low probability) and start 6 s into the window: gap > threshold ->
    # nothing of the window is kept and seek = previous + gap
    toks = [TSB + 300, 10, 11, 12, TSB + 320, TSB + 320, 13, TSB + 400, TSB + 400]
    bad = [(6.00, 6.01, 0.05), (6.01, 6.02, 0.05), (6.02, 6.03, 0.05), (6.5, 7.5, 0.9)]
    good = [(0.2, 0.9, 0.9), (1.0, 1.8, 0.9), (1.9, 2.6, 0.9), (3.0, 3.9, 0.9)]
    res, calls, aligns = _run_words([DecodingResult(toks, -0.3, 0.01)], [bad, good], n_samples=160 * 1500, hallucination_silence_threshold=2.0)
    assert len(calls) >= 2
    assert all(s.start >= 6.0 - 1e-3 for s in res.segments)                    # window 1 contributed nothing; window 2 starts at 6.0 s
    assert calls[1][0] == []                                                   # and no tokens of window 1 condition window 2


def test_anomalous_segment_between_silences_is_dropped():
    # segment 2 (13) is an anomaly with > threshold of silence before and after -> it and what follows are removed, seek -> its start
    toks = [TSB, 10, 11, TSB + 100, TSB + 100 + 250, 13, TSB + 100 + 260, TSB + 800, 14, TSB + 900, TSB + 900]
    toks = [TSB, 10, 11, TSB + 100, TSB + 350, 13, TSB + 360, TSB + 360]
    w = [(0.1, 0.9, 0.9), (1.0, 1.9, 0.9), (7.0, 7.01, 0.05)]
    res, calls, aligns = _run_words([DecodingResult(toks, -0.3, 0.01)], [w], n_samples=160 * 2000, hallucination_silence_threshold=2.0)
    first_window = [s for s in res.segments if s.start < 7.0]
    assert [s.tokens for s in first_window] == [toks[:4]]                      # the anomalous second segment is gone
    assert len(calls) >= 2


def test_word_mode_final_filters():
    # a segment with text longer than 10 characters but no aligned words is dropped (alignment failed); a segment whose words are
    # anomalous is dropped too
    toks = [TSB, 10, 11, 12, TSB + 100, TSB + 100]
    calls = []

    def decode_fn(mels, prompts, temps, uniforms):
        calls.append(1)
        return [DecodingResult(toks, -0.3, 0.01)]

    mel = np.zeros((160 * 500 // 160 + 3000, 80), np.float32)
    res = T.transcribe_batch([mel], [160 * 500], decode_fn, FakeTok(), ST, word_timestamps=True, align_fn=lambda m, t, f: [[]])[0]
    assert res.segments == []                                                  # " w10 w11 w12" has 12 characters and no words
    bad = lambda m, t, f: [[WordTiming(" w%d" % x, [x], 0.5 + 0.01 * k, 0.5 + 0.01 * k + 0.005, 0.05) for k, x in enumerate(t[0])]]
    res = T.transcribe_batch([mel], [160 * 500], decode_fn, FakeTok(), ST, word_timestamps=True, align_fn=bad)[0]
    assert res.segments == []


def test_language_detection_feeds_per_clip_languages():
    seen = {}

    def detect_fn(mels):
        seen["detect"] = mels.shape
        return [(7, 0.9), (3, 0.8)]

    def decode_fn(mels, prompts, temps, uniforms, langs):
        seen.setdefault("langs", []).append(list(langs))
        return [DecodingResult([TSB, 10, TSB + 1400, TSB + 1400], -0.3, 0.01) for _ in range(mels.shape[0])]

    mels = [np.zeros((160 * 1000 // 160 + 3000, 80), np.float32), np.zeros((160 * 4000 // 160 + 3000, 80), np.float32)]
    res = T.transcribe_batch(mels, [160 * 1000, 160 * 4000], decode_fn, FakeTok(), ST, language=None, detect_fn=detect_fn)
    assert seen["detect"] == (2, 3000, 80)
    assert seen["langs"][0] == [7, 3] and all(l == [3] for l in seen["langs"][1:])      # clip 0 finishes first; clip 1 keeps its language
    assert [r.language for r in res] == [7, 3]
