"""CPU tests of the text-side token rules (tokenizer.py): the tiktoken file format, byte-pair merging by rank with the GPT-2 split
pattern, nonSpeechTokens / the suppress list (WhisperTokenizer.swift:186-217,489-532; WhisperDecoding.swift:190-206).  No real
vocabulary exists offline: a synthetic one in the same format, with hand-checkable merges."""
import base64

import pytest

from mlx_swift_audio_amd import tokenizer as TK


def _vocab():
    """256 byte tokens (ranks 0..255) + a few merges with chosen ranks."""
    toks = [bytes([i]) for i in range(256)]
    merges = [b" -", b" '", b"<<", b"<<<", b" <<", b"((", b" ((", b"he", b"ll", b"hell", b"hello", b" hello", b" w", b"or", b" wor",
              "♪".encode(), " ♪".encode(), "♪♪".encode(), b"[[", b" (", b" \xe2"]   # (" \xe2": like GPT-2's "Ġâ", a space merged with a lead byte)
    toks += merges
    lines = [base64.b64encode(t).decode() + " " + str(i) for i, t in enumerate(toks)]
    return "\n".join(lines) + "\n\n", {t: i for i, t in enumerate(toks)}


def test_parse_and_roundtrip():
    text, want = _vocab()
    ranks = TK.parse_tiktoken_bpe(text)
    assert ranks == want
    with pytest.raises(ValueError):
        TK.parse_tiktoken_bpe("bm90 a-rank-less-line-without-number x y\nZZ")
    bpe = TK.BPE(ranks, {"<|endoftext|>": 300})
    ids = bpe.encode_ordinary("hello world")
    # "hello" is one token; " world" -> " wor" + "l" + "d" (no "ld" merge in this vocabulary)
    assert ids == [want[b"hello"], want[b" wor"], ord("l"), ord("d")]
    assert bpe.decode(ids) == "hello world" and bpe.decode([300]) == "<|endoftext|>"
    # lowest rank wins, not leftmost: in "hell" the pair "ll" (rank 264) is merged after "he" (rank 263), then "hell" exists
    assert bpe.encode_ordinary("hell") == [want[b"hell"]]
    # the split pattern keeps the leading space with the word and isolates punctuation runs
    assert bpe.encode_ordinary(" hello((") == [want[b" hello"], want[b"(("]]
    assert bpe.decode(bpe.encode_ordinary("naïve ♪ ok")) == "naïve ♪ ok"           # multi-byte characters fall back to bytes


def test_non_speech_and_suppress_lists():
    text, want = _vocab()
    bpe = TK.BPE(TK.parse_tiktoken_bpe(text))
    ns = TK.non_speech_tokens(bpe.encode_ordinary)
    assert ns == sorted(set(ns))
    # single-character symbols are single byte tokens here; their " x" forms are two tokens unless a merge exists
    for ch in "\"#()*+/:;<=>@[\\]^_`{|}~":
        assert ord(ch) in ns
    assert want[b" -"] in ns and want[b" '"] in ns                                  # the word-start hyphen / quote
    assert want[b"<<"] in ns and want[b"<<<"] in ns and want[b" <<"] in ns and want[b"(("] in ns and want[b" (("] in ns and want[b"[["] in ns
    assert want[b" ("] in ns
    # musical symbols: first token even when the symbol needs several; "♪" / " ♪" are merged tokens here, the others start with byte 0xE2
    assert want["♪".encode()] in ns and want[" ♪".encode()] in ns and 0xE2 in ns and want[b" \xe2"] in ns
    assert want["♪♪".encode()] in ns                                                 # one token -> counted; "♪♪♪" is two tokens -> not
    assert ord("-") not in ns and ord("'") not in ns and ord("h") not in ns and ord(" ") not in ns      # plain "-" / "'" stay legal

    class Sp:
        transcribe, translate, sot, sot_prev, sot_lm, no_speech = 50359, 50358, 50258, 50361, 50360, 50362
    sup = TK.suppress_tokens(bpe.encode_ordinary, Sp)
    assert set(ns) < set(sup) and {50359, 50358, 50258, 50361, 50360, 50362} <= set(sup) and sup == sorted(sup)
    assert TK.blank_tokens(bpe.encode_ordinary) == [ord(" ")]
