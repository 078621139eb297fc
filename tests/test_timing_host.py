"""CPU tests of the word-timestamp back half (timing.py: mergePunctuations, duration clipping, addWordTimestamps, anomaly scoring):
host text / float logic of WhisperTiming.swift:311-556, 847-1060 and the word splitter of WhisperTokenizer.swift:546-670.
Expected values are worked out by hand from the Swift's rules (the reference ships no fixtures for them)."""
from dataclasses import dataclass, field

import pytest

from mlx_swift_audio_amd import timing as T

WT = T.WordTiming
EOT = 1000
VOCAB = {1: " Hello", 2: ",", 3: " wor", 4: "ld", 5: " \"", 6: "quoted", 7: "\"", 8: ".", 9: " (", 10: "x", 11: ")", 12: " -", 13: "dash",
         20: "你", 21: "好", 22: "世界", 30: "Start"}


def decode(tokens):
    return "".join(VOCAB[t] for t in tokens)


def test_split_to_word_tokens_whitespace_and_character_level():
    split = T.make_split_to_word_tokens(decode, EOT)
    words, groups = split([1, 2, 3, 4, 8, EOT])
    assert words == [" Hello,", " world.", ""] and groups == [[1, 2], [3, 4, 8], [EOT]]
    # no leading-space token at the start still opens the first word; a missing EOT closes the last word at the end
    words, groups = split([30, 3, 4])
    assert words == ["Start", " world"] and groups == [[30], [3, 4]]
    # more than half of the characters from a space-less script -> one "word" per token
    words, groups = split([20, 21, 22, EOT])
    assert words == ["你", "好", "世界", ""] and groups == [[20], [21], [22], [EOT]]
    assert split([EOT]) == ([""], [[EOT]]) and split([]) == ([], [])


def test_merge_punctuations_prepend_and_append():
    # tokens: ' "' quoted '"' '.'  ' (' x ')'  -> ' "quoted".'  ' (x)'
    a = [WT(' "', [5], 0.0, 0.1, 1), WT("quoted", [6], 0.1, 0.5, 1), WT('"', [7], 0.5, 0.6, 1), WT(".", [8], 0.6, 0.7, 1),
         WT(" (", [9], 0.7, 0.8, 1), WT("x", [10], 0.8, 0.9, 1), WT(")", [11], 0.9, 1.0, 1)]
    T.merge_punctuations(a)
    assert [w.word for w in a] == [' "quoted".', " (x)"]
    assert [w.tokens for w in a] == [[5, 6, 7, 8], [9, 10, 11]]
    assert (a[0].start, a[0].end, a[1].start, a[1].end) == (0.0, 0.7, 0.7, 1.0)
    # a punctuation word WITHOUT the leading space is not prepended; a previous word ending in a space blocks appending
    b = [WT("-", [12], 0.0, 0.1, 1), WT("dash", [13], 0.1, 0.3, 1), WT("end ", [30], 0.3, 0.4, 1), WT(".", [8], 0.4, 0.5, 1)]
    T.merge_punctuations(b)
    assert [w.word for w in b] == ["-", "dash", "end ", "."]
    one = [WT(".", [8], 0, 1, 1)]
    T.merge_punctuations(one)
    assert len(one) == 1


def test_duration_thresholds_and_sentence_clipping():
    a = [WT("a", [1], 0.0, 0.2, 1), WT("b", [1], 0.2, 0.2, 1), WT("c", [1], 0.2, 0.6, 1), WT("d", [1], 0.6, 1.6, 1)]   # durations .2, 0, .4, 1.0
    med, mx = T.calculate_duration_thresholds(a)
    assert med == pytest.approx(0.4) and mx == pytest.approx(0.8)               # zero durations are dropped, odd count -> middle
    a2 = a + [WT("e", [1], 1.6, 3.6, 1)]                                      # .2 .4 1.0 2.0 -> even count: (0.4 + 1.0) / 2 = 0.7 cap
    med2, mx2 = T.calculate_duration_thresholds(a2)
    assert med2 == pytest.approx(0.7) and mx2 == pytest.approx(1.4)
    assert T.calculate_duration_thresholds([WT("z", [1], 1, 1, 1)]) == (0.0, 0.0)
    # a long word that IS a sentence mark is cut at its end; a long word AFTER a sentence mark is cut at its start
    s = [WT("hi", [1], 0.0, 0.3, 1), WT(".", [8], 0.3, 2.3, 1), WT("next", [1], 2.3, 5.3, 1), WT("ok", [1], 5.3, 5.5, 1)]
    T.clip_at_sentence_boundaries(s, 0.8)
    assert s[1].end == pytest.approx(1.1) and s[1].start == pytest.approx(0.3)
    assert s[2].start == pytest.approx(4.5) and s[2].end == pytest.approx(5.3)
    assert (s[3].start, s[3].end) == (5.3, 5.5)


def test_segment_boundary_clipping_after_a_pause():
    # pause before the first word (its END is 9 s after the last speech) > 4 x median; first word too long, second one too
    w = [WT("a", [], 10.0, 14.0, 1), WT("b", [], 14.0, 17.0, 1), WT("c", [], 17.0, 17.2, 1)]
    T.clip_at_segment_boundaries(w, last_speech_timestamp=5.0, median_duration=0.5, max_duration=1.0)
    # second word longer than max: boundary = max(17 / 2, 17 - 1) = 16 -> first.end = second.start = 16; first.start = max(0, 16 - 1)
    assert (w[0].start, w[0].end, w[1].start, w[1].end) == (15.0, 16.0, 16.0, 17.0)
    q = [WT("a", [], 10.0, 10.4, 1)]
    T.clip_at_segment_boundaries(q, last_speech_timestamp=9.9, median_duration=0.5, max_duration=1.0)   # no long pause: untouched
    assert (q[0].start, q[0].end) == (10.0, 10.4)


@dataclass
class Seg:
    tokens: list
    start: float
    end: float
    words: list = field(default_factory=list)


def test_add_word_timestamps_distributes_and_adjusts():
    # two segments: [<|0.00|> Hello , wor ld <|2.00|>] and [<|2.00|> Start . <|3.00|>]; timestamp tokens are >= EOT
    segs = [Seg([EOT + 10, 1, 2, 3, 4, EOT + 110], 0.0, 2.0), Seg([EOT + 110, 30, 8, EOT + 160], 2.0, 3.0)]
    align = [WT(" Hello", [1], 0.10, 0.50, 0.9), WT(",", [2], 0.50, 0.55, 0.8), WT(" world", [3, 4], 0.55, 1.20, 0.7),
             WT("Start", [30], 2.10, 2.60, 0.6), WT(".", [8], 2.60, 2.70, 0.5)]
    last = T.add_word_timestamps(segs, align, EOT, time_offset=30.0, last_speech_timestamp=29.0)
    assert [w.word for w in segs[0].words] == [" Hello,", " world"]
    assert [w.word for w in segs[1].words] == ["Start."]
    assert segs[0].words[0].start == pytest.approx(30.10) and segs[0].words[0].end == pytest.approx(30.55)
    assert segs[0].words[1].end == pytest.approx(31.20)
    # segment bounds follow the first / last word (the segment-level stamps are not >0.5 s inside the words)
    assert segs[0].start == pytest.approx(30.10) and segs[0].end == pytest.approx(31.20)
    assert segs[1].start == pytest.approx(32.10) and segs[1].end == pytest.approx(32.70)
    assert last == pytest.approx(32.70)
    # a first word that starts more than 0.5 s BEFORE the segment's own start stamp is pulled in to the segment start
    segs2 = [Seg([1, 3, 4], 31.0, 33.0)]
    al2 = [WT(" Hello", [1], 0.2, 1.6, 0.9), WT(" world", [3, 4], 1.6, 2.0, 0.9)]          # with offset 30: starts at 30.2 < 31.0 - 0.5
    T.add_word_timestamps(segs2, al2, EOT, time_offset=30.0, last_speech_timestamp=30.0)
    med, _ = T.calculate_duration_thresholds([WT("", [], 0.2, 1.6, 1), WT("", [], 1.6, 2.0, 1)])
    assert segs2[0].start == pytest.approx(max(0.0, min(31.6 - med, 31.0))) and segs2[0].words[0].start == pytest.approx(segs2[0].start)
    assert T.add_word_timestamps([], al2, EOT, 0.0, 7.0) == 7.0
    assert T.add_word_timestamps([Seg([EOT + 1], 0, 1)], al2, EOT, 0.0, 7.0) == 7.0   # no text tokens


def test_anomaly_scores():
    assert T.word_anomaly_score(WT("a", [], 0.0, 0.5, 0.9)) == 0.0
    assert T.word_anomaly_score(WT("a", [], 0.0, 0.5, 0.1)) == 1.0                          # low probability
    assert T.word_anomaly_score(WT("a", [], 0.0, 0.033, 0.9)) == pytest.approx(1.5)         # (0.133 - 0.033) * 15
    assert T.word_anomaly_score(WT("a", [], 0.0, 3.5, 0.9)) == pytest.approx(1.5)           # 3.5 - 2
    ok = [WT("w%d" % i, [], i, i + 0.4, 0.9) for i in range(10)]
    assert not T.is_segment_anomaly(ok) and not T.is_segment_anomaly([]) and not T.is_segment_anomaly(None)
    bad = [WT("w", [], 0.0, 0.033, 0.1), WT("v", [], 1.0, 1.01, 0.9)] + ok                   # 2.5 + 1.845 -> >= 3
    assert T.is_segment_anomaly(bad)
    # single-character punctuation words are skipped ("..." is NOT a substring of string.punctuation, so it counts)
    assert not T.is_segment_anomaly([WT(".", [], 0, 0, 0.0), WT(",", [], 0, 0, 0.0)])
    assert T.is_segment_anomaly([WT("...", [], 0.0, 0.0, 0.0)])                             # score 1 + 1.995 >= 1 word
    # "almost all words anomalous": 2 words, each exactly 1.0 -> 2.0 + 0.01 >= 2
    assert T.is_segment_anomaly([WT("a", [], 0, 0.5, 0.1), WT("b", [], 1, 1.5, 0.1)])
