"""GPU parity: S3Tokenizer (HIP fp32, exact-length per clip) vs the fp32 CPU oracle that follows the Swift's padded-batch +
mask arithmetic.  Token ids (integers) must be bit-exact; a base-3 digit may differ ONLY where the oracle's own pre-round FSQ
value tanh(.)*0.999 lies within BOUNDARY_TOL of a rounding boundary (+-0.5), which the test proves digit by digit."""
import numpy as np
import pytest

from mlx_swift_audio_amd import synthetic as S
from oracle import logmel as OL
from oracle import s3tok as OS

pytestmark = pytest.mark.gpu

BOUNDARY_TOL = 2e-3


def assert_ids_exact_or_on_boundary(got, ref, pre, what=""):
    """got / ref: int ids [T]; pre: oracle pre-round values [T, 8].  Every digit that differs must sit on a rounding boundary."""
    bad = 0
    for t in np.nonzero(np.asarray(got) != np.asarray(ref))[0]:
        bad += 1
        for i in range(8):
            dg, dr = (int(got[t]) // 3 ** i) % 3, (int(ref[t]) // 3 ** i) % 3
            if dg != dr:
                dist = abs(abs(float(pre[t, i])) - 0.5)
                assert abs(dg - dr) == 1 and dist < BOUNDARY_TOL, f"{what} t={t} digit {i}: {dg} vs {dr}, pre-round {pre[t, i]} is {dist} from +-0.5"
    return bad


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
    ref, rn, pre = ora.quantize(mel, np.asarray(mel_len))
    np.testing.assert_array_equal(n, rn)
    for b in range(3):
        assert_ids_exact_or_on_boundary(got[b, :n[b]], ref[b, :n[b]], pre[b], f"clip {b}")
        assert np.all(got[b, n[b]:] == 0)
    tok.close()


def test_long_audio_windows_and_merge(ctx):
    cfg, tok, ora = _prepare(ctx, "s3_micro", 3)
    rng = np.random.default_rng(0)
    L = 3000 + 2600 + 700                                        # three windows
    mel = (0.5 * rng.standard_normal((1, cfg.n_mels, L))).astype(np.float32)
    got, n = tok.quantize(mel, [L])
    segs, pres, start = [], [], 0
    while start < L:
        end = min(start + 3000, L)
        ids, ln, pre = ora.quantize(mel[:, :, start:end], np.asarray([end - start]))
        segs.append(ids[0, :ln[0]].tolist())
        pres.append([pre[0, t] for t in range(ln[0])])
        if end == L:
            break
        start += 2600
    ref = OS.merge_tokenized_segments(segs)
    ref_pre = np.asarray(OS.merge_tokenized_segments(pres))       # the same slicing applied to the pre-round rows
    assert n[0] == len(ref)
    assert_ids_exact_or_on_boundary(got[0, :n[0]], np.asarray(ref), ref_pre, "long audio")
    tok.close()


@pytest.mark.parametrize("cfg_name", ["s3_v2", "s3_v3"])
def test_real_geometry_tokens_match_oracle(ctx, cfg_name):
    """S3TokenizerV2 / V3 at their real width and depth (d 1280, 20 heads, 6 / 12 FSMN blocks) on a few seconds of input."""
    from mlx_swift_audio_amd import audio as A
    cfg, tok, ora = _prepare(ctx, cfg_name, 4)
    clips = [OL.synth_clip(5, 16000 * 3), OL.synth_clip(6, 16000 * 2 + 777)]
    mels = [A.s3_log_mel_spectrogram(ctx, c, cfg.n_mels) for c in clips]
    T = max(m.shape[1] for m in mels)
    mel = np.zeros((2, cfg.n_mels, T), np.float32)
    for b, m in enumerate(mels):
        mel[b, :, :m.shape[1]] = m
    lens = [m.shape[1] for m in mels]
    got, n = tok.quantize(mel, lens)
    ref, rn, pre = ora.quantize(mel, np.asarray(lens))
    np.testing.assert_array_equal(n, rn)
    bad = sum(assert_ids_exact_or_on_boundary(got[b, :n[b]], ref[b, :n[b]], pre[b], f"{cfg_name} clip {b}") for b in range(2))
    assert bad <= 0.05 * int(n.sum()), (bad, int(n.sum()))       # boundary cases are rare by construction
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
