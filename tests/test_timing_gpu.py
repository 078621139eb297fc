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
