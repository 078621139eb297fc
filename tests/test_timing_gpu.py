"""GPU parity of the word-timestamp alignment (section 8f rank 3): mia_whisper_align vs the oracle's restatement of findAlignment."""
import numpy as np
import pytest

from oracle import whisper as OW

pytestmark = pytest.mark.gpu


def _setup(ctx, dtype_name="f16"):
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import whisper as HW
    dims = OW.DIMS["micro"]
    w = OW.synthetic_weights(dims, seed=11, round_to=dtype_name)
    return dims, OW.WhisperOracle(dims, w), HW.WhisperModel.load(ctx, dims, w, m.F16 if dtype_name == "f16" else m.BF16)


def test_alignment_matrix_path_and_probs(ctx):
    from mlx_swift_audio_amd import timing as HT
    dims, ora, model = _setup(ctx)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    rng = np.random.default_rng(3)
    mel = OW.round_array((0.5 * rng.standard_normal((2, 2 * dims.n_audio_ctx, dims.n_mels))).astype(np.float32), "f16")
    model.encode(mel)
    xa = ora.encode(mel)
    sot = st.sot_sequence(3, "transcribe")
    texts = [rng.integers(300, 5000, 9).tolist(), rng.integers(300, 5000, 5).tolist()]       # ragged
    frames = [2 * dims.n_audio_ctx, 2 * dims.n_audio_ctx - 10]
    heads = [(0, 1), (dims.n_text_layer - 1, 0), (dims.n_text_layer - 1, dims.n_text_head - 1)]
    seqs = [sot + [st.no_timestamps] + t + [st.eot] for t in texts]
    probs, paths, mat = HT.align(model, seqs, heads, frames, len(sot), st.eot, want_matrix=True)
    for b in range(2):
        want_mat, want_probs = OW.alignment_matrix(ora, xa[b:b + 1], seqs[b], heads, frames[b], st.eot)
        n, F = len(seqs[b]), frames[b] // 2
        got = mat[b, :n, :F]
        # standardised attention weights are O(1); f16 q/k and the median's selection make this a loose elementwise bound
        assert np.abs(got - want_mat).mean() < 0.02 and np.abs(got - want_mat).max() < 0.5
        np.testing.assert_allclose(probs[b, :n - 1], want_probs, rtol=0.08, atol=2e-5)
        # the device path must be the exact DTW of the device matrix (host-side algorithm, tie rules included)
        ti, tj = OW.dtw(-got[len(sot):n - 1])
        assert (ti, tj) == paths[b]
        assert tj[0] == 0 and tj[-1] == F - 1 and ti[0] == 0 and ti[-1] == n - 2 - len(sot)
    model.close()


def test_find_alignment_word_times(ctx):
    from mlx_swift_audio_amd import timing as HT
    dims, ora, model = _setup(ctx)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    rng = np.random.default_rng(4)
    mel = OW.round_array((0.5 * rng.standard_normal((1, 2 * dims.n_audio_ctx, dims.n_mels))).astype(np.float32), "f16")
    model.encode(mel)
    text = rng.integers(300, 5000, 7).tolist()
    sot = st.sot_sequence(0, "transcribe")

    def split(tokens):           # synthetic tokenizer: words of 2, 3, 2 tokens + the eot group
        groups = [tokens[0:2], tokens[2:5], tokens[5:7], tokens[7:]]
        return ["w0", "w1", "w2", ""], groups

    heads = [(dims.n_text_layer - 1, 0), (dims.n_text_layer - 1, 1)]
    res = HT.find_alignment(model, [text], [2 * dims.n_audio_ctx], sot, st, heads, split)[0]
    assert [w.word for w in res] == ["w0", "w1", "w2"] and [w.tokens for w in res] == [text[0:2], text[2:5], text[5:7]]
    seq = sot + [st.no_timestamps] + text + [st.eot]
    _, paths, _ = HT.align(model, [seq], heads, [2 * dims.n_audio_ctx], len(sot), st.eot)
    want = OW.word_times(paths[0][0], paths[0][1], split(text + [st.eot])[1])
    for w, (s, e) in zip(res, want):
        assert w.start == pytest.approx(s) and w.end == pytest.approx(e) and 0.0 <= w.probability <= 1.0
    assert all(res[i].start <= res[i + 1].start for i in range(len(res) - 1))
    model.close()


def test_align_errors(ctx):
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import timing as HT
    dims, ora, model = _setup(ctx)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    with pytest.raises(m.MiaError):
        HT.align(model, [[st.sot, st.no_timestamps, 500, st.eot]], [(0, 0)], [100], 1, st.eot)        # no encode yet
    mel = np.zeros((1, 2 * dims.n_audio_ctx, dims.n_mels), np.float32)
    model.encode(mel)
    with pytest.raises(m.MiaError):
        HT.align(model, [[st.sot, st.no_timestamps, 500, st.eot]], [(99, 0)], [100], 1, st.eot)       # head out of range
    model.close()


def test_transcribe_word_timestamps_and_language_detection(ctx):
    """WhisperSTT.transcribe(timestamps: .word, language: nil) on the device path (WhisperSTT.swift:155-161, 440-590): the language is
    detected per clip on its first window (equal to the oracle's detectLanguage) and used for that clip's sot sequence; ONE alignment
    call per window (encoder + teacher-forced pass + DTW on the GPU) gives the words, whose times equal the oracle's alignment of
    the same tokens to within a frame; add_word_timestamps deals them to the kept segments and seek follows the last word.
    (A random-init model aligns at random with near-zero word probabilities, which the anomaly rules would discard wholesale: the
    test's align wrapper records the device alignment for the comparison and hands the loop the same words with probability 0.9.)"""
    import dataclasses
    from mlx_swift_audio_amd import transcribe as TR
    from mlx_swift_audio_amd import whisper as HW
    from oracle import logmel as OLM
    import mlx_swift_audio_amd as m
    dims = OW.DIMS["micro"]
    w = OW.synthetic_weights(dims, seed=157, style="peaky", round_to="f16")        # non-degenerate decoding (tests/test_whisper_steps_gpu.py)
    ora = OW.WhisperOracle(dims, w)
    model = HW.WhisperModel.load(ctx, dims, w, m.F16)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    sup = OW.synthetic_suppress_list(st)

    class Tok:
        def decode(self, toks):
            return "".join(" w%d" % t for t in toks)

    def split(tokens):                                   # synthetic tokenizer: every text token is a word, the eot its own group
        return [" w%d" % t for t in tokens[:-1]] + [""], [[t] for t in tokens]

    heads = [(dims.n_text_layer - 1, 0), (dims.n_text_layer - 1, 1)]
    win = dims.n_audio_ctx * 2 * 160                     # a 2 s window at the micro size
    clips = [OLM.synth_clip(0, int(win * 1.6)), OLM.synth_clip(1, int(win * 0.7))]
    stt = TR.WhisperSTT(ctx, model, Tok(), sup, [220], alignment_heads=heads, split_to_word_tokens=split)
    recorded = []
    dev_align = stt._align_fn(None)

    def align_fn(mels, text_tokens, num_frames, langs):
        out = dev_align(mels, text_tokens, num_frames, langs)
        recorded.append((mels.copy(), [list(t) for t in text_tokens], list(num_frames), list(langs), [[dataclasses.replace(x) for x in wt] for wt in out]))
        return [[dataclasses.replace(x, probability=0.9) for x in wt] for wt in out]

    mels = [OLM.whisper_log_mel_spectrogram(c, dims.n_mels, padding=TR.N_SAMPLES) for c in clips]
    got = TR.transcribe_batch(mels, [c.shape[0] for c in clips], stt._decode_fn(None, True, 8), Tok(), st, language=None, max_tokens=8,
                              rng=np.random.default_rng(3), n_audio_ctx=dims.n_audio_ctx, word_timestamps=True, align_fn=align_fn,
                              detect_fn=stt._detect_fn(), logprob_threshold=-20.0, compression_ratio_threshold=50.0, no_speech_threshold=None)
    first = np.stack([TR.pad_or_trim_mel(x[:2 * dims.n_audio_ctx], 2 * dims.n_audio_ctx) for x in mels])
    xa0 = ora.encode(OW.round_array(first, "f16"))
    for b, r in enumerate(got):
        li, _ = ora.detect_language(xa0[b:b + 1], st)
        assert r.language == li                          # per-clip language, the oracle's argmax
        assert r.passes >= 1
    assert recorded, "no alignment call happened"
    n_words = 0
    for mel_w, texts, frames, langs, outs in recorded:
        xa = ora.encode(OW.round_array(mel_w, "f16"))
        for b, (text, wt) in enumerate(zip(texts, outs)):
            if not wt:                                   # findAlignment's guard: a window shorter than two encoder frames has no alignment
                assert frames[b] // 2 < 2, (frames[b], text)
                continue
            assert [x.word for x in wt] == [" w%d" % t for t in text] and [x.tokens for x in wt] == [[t] for t in text]
            sot = st.sot_sequence(langs[b], "transcribe")
            seq = sot + [st.no_timestamps] + text + [st.eot]
            mat, _ = OW.alignment_matrix(ora, xa[b:b + 1], seq, heads, frames[b], st.eot)
            ti, tj = OW.dtw(-mat[len(sot):len(seq) - 1])
            want = OW.word_times(ti, tj, [[t] for t in text] + [[st.eot]])
            for x, (s0, e0) in zip(wt, want):
                assert abs(x.start - s0) <= 0.0401 and abs(x.end - e0) <= 0.0401, (x, s0, e0)      # one 20 ms frame each way
                assert 0.0 <= x.start <= x.end <= frames[b] / 100.0 + 1e-6
            n_words += len(wt)
    assert n_words >= 4
    kept = [s for r in got for s in r.segments]
    assert kept and all(s.words for s in kept)
    for s in kept:
        assert [x.word for x in s.words] == [" w%d" % t for t in s.tokens if t < st.eot]
        assert all(a.end <= b.start + 1e-6 or a.end <= b.end for a, b in zip(s.words, s.words[1:]))
    model.close()
