"""GPU parity: S3Tokenizer (HIP fp32, exact-length per clip) vs the fp32 CPU oracle that follows the Swift's padded-batch +
mask arithmetic.  Token ids must be bit-exact wherever the oracle's pre-round FSQ value tanh(.)*0.999 is farther than 2e-3
from a rounding boundary (+-0.5); the test asserts the match rate and that every mismatch sits on such a boundary."""
import numpy as np
import pytest

from mlx_swift_audio_amd import synthetic as S
from oracle import logmel as OL
from oracle import s3tok as OS

pytestmark = pytest.mark.gpu


def _prepare(ctx, cfg_name, seed):
    from mlx_swift_audio_amd import s3tok as HS
    cfg = S.S3_CONFIGS[cfg_name]
    w = S.s3_weights(cfg, seed)
    return cfg, HS.S3Tokenizer.load(ctx, cfg, w), OS.S3Oracle(cfg, w)


def test_tokens_match_oracle_ragged_batch(ctx):
    from mlx_swift_audio_amd import audio as A
    cfg, tok, ora = _prepare(ctx, "s3_micro", 2)
    lens = [1600 * 4, 1600 * 7 + 333, 16000 * 3]
    clips = [OL.synth_clip(i, n) for i, n in enumerate(lens)]
    T = max(lens) // 160
    mel = np.zeros((3, cfg.n_mels, T), np.float32)
    mel_len = []
    for b, c in enumerate(clips):
        m = A.s3_log_mel_spectrogram(ctx, c, cfg.n_mels)         # [n_mels, frames] from the HIP front end
        mel[b, :, :m.shape[1]] = m
        mel_len.append(m.shape[1])
    got, n = tok.quantize(mel, mel_len)
    ref, rn, h = ora.quantize(mel, np.asarray(mel_len))
    np.testing.assert_array_equal(n, rn)
    total = bad = 0
    for b in range(3):
        for t in range(n[b]):
            total += 1
            if got[b, t] != ref[b, t]:
                bad += 1
                # the pre-round values of the differing digits must sit on a rounding boundary
                digits_g = [(got[b, t] // 3 ** i) % 3 for i in range(8)]
                digits_r = [(ref[b, t] // 3 ** i) % 3 for i in range(8)]
                assert sum(a != c for a, c in zip(digits_g, digits_r)) <= 2
        assert np.all(got[b, n[b]:] == 0)
    assert bad / total <= 0.02, (bad, total)
    tok.close()


def test_long_audio_windows_and_merge(ctx):
    cfg, tok, ora = _prepare(ctx, "s3_micro", 3)
    rng = np.random.default_rng(0)
    L = 3000 + 2600 + 700                                        # three windows
    mel = (0.5 * rng.standard_normal((1, cfg.n_mels, L))).astype(np.float32)
    got, n = tok.quantize(mel, [L])
    segs, start = [], 0
    while start < L:
        end = min(start + 3000, L)
        ids, ln, _ = ora.quantize(mel[:, :, start:end], np.asarray([end - start]))
        segs.append(ids[0, :ln[0]].tolist())
        if end == L:
            break
        start += 2600
    ref = OS.merge_tokenized_segments(segs)
    assert n[0] == len(ref)
    match = np.mean(np.asarray(ref) == got[0, :n[0]])
    assert match >= 0.98, match
    tok.close()


def test_s3_error_paths(ctx):
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import s3tok as HS
    cfg = S.S3_CONFIGS["s3_micro"]
    w = S.s3_weights(cfg, 1)
    bad = dict(w)
    del bad["encoder.blocks.0.attn.fsmn_block.weight"]
    with pytest.raises(m.MiaError):
        HS.S3Tokenizer.load(ctx, cfg, bad)
