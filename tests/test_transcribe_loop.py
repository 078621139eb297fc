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
