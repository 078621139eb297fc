"""GPU parity: HIP log-mel (through the C ABI) vs the CPU oracle on the same seeded inputs.

Tolerance: |hip - oracle| <= 1e-3 absolute in normalised log-mel units ((log10+4)/4) for fp32 output;
observed error on broadband signals is ~1e-6 (both sides are fp32 DFTs of 400 points).  The loose bound covers
bins near the max-8 clamp floor where an fp32 DFT has large *relative* error.  bf16/f16 outputs are compared
after rounding the oracle the same way (<= 1 ulp of the 16-bit type).
"""
import numpy as np
import pytest

from oracle import logmel as O

pytestmark = pytest.mark.gpu

TOL = 1e-3


def _signals():
    rng = np.random.default_rng(7)
    t = np.arange(32000) / 16000.0
    sweep = (0.5 * np.sin(2 * np.pi * (100 + 1500 * t) * t)).astype(np.float32)
    noise = (0.1 * rng.standard_normal(24000)).astype(np.float32)
    click = np.zeros(16000, np.float32)
    click[8000] = 1.0
    return {"sweep": sweep, "noise": noise, "click": click}


@pytest.mark.parametrize("n_mels", [80, 128])
@pytest.mark.parametrize("name", ["sweep", "noise", "click"])
def test_whisper_logmel_matches_oracle(ctx, name, n_mels):
    from mlx_swift_audio_amd import audio as A
    x = _signals()[name]
    got = A.whisper_log_mel_spectrogram(ctx, x, n_mels)
    ref = O.whisper_log_mel_spectrogram(x, n_mels)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, atol=TOL, rtol=0)


def test_whisper_logmel_padding_and_truncated_output(ctx):
    """As called by transcribe (WhisperSTT.swift:140-145): +30 s of zeros; we emit the first 3000 frames only,
    but the clamp floor must still come from the whole padded utterance."""
    from mlx_swift_audio_amd import audio as A
    x = O.synth_clip(3, 160000)                      # BASELINE config 0: one 10 s clip
    ref = O.whisper_log_mel_spectrogram(x, 80, padding=480000)
    got = A.whisper_log_mel_spectrogram(ctx, x, 80, padding=480000, n_frames=3000)
    np.testing.assert_allclose(got, ref[:3000], atol=TOL, rtol=0)
    full = A.whisper_log_mel_spectrogram(ctx, x, 80, padding=480000)
    assert full.shape == (4000, 80)
    np.testing.assert_allclose(full, ref, atol=TOL, rtol=0)


def test_frames_past_utterance_are_zero(ctx):
    """padOrTrimMel (WhisperSTT.swift:624-635) pads the window with 0.0 in normalised units."""
    from mlx_swift_audio_amd import audio as A
    x = O.synth_clip(1, 16000)
    got = A.whisper_log_mel_spectrogram(ctx, x, 128, padding=0, n_frames=300)
    ref = O.whisper_log_mel_spectrogram(x, 128)
    np.testing.assert_allclose(got[:100], ref, atol=TOL, rtol=0)
    assert np.all(got[100:] == 0.0)


def test_ragged_batch(ctx):
    from mlx_swift_audio_amd import audio as A
    clips = [O.synth_clip(i, n) for i, n in enumerate([16000, 4321, 48000, 801])]
    got = A.whisper_log_mel_spectrogram(ctx, clips, 128, padding=8000, n_frames=350)
    assert got.shape == (4, 350, 128)
    for b, c in enumerate(clips):
        ref = O.whisper_log_mel_spectrogram(c, 128, padding=8000)
        n = ref.shape[0]
        np.testing.assert_allclose(got[b, :n], ref, atol=TOL, rtol=0)
        assert np.all(got[b, n:] == 0.0)


@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
def test_16bit_outputs(ctx, dtype_name):
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import audio as A
    x = O.synth_clip(2, 32000)
    ref = O.whisper_log_mel_spectrogram(x, 128)
    if dtype_name == "bf16":
        got = A.bf16_to_f32(A.whisper_log_mel_spectrogram(ctx, x, 128, dtype=m.BF16))
        ulp = 2.0 ** -7      # values are in ~[-1.5, 2): 1 bf16 ulp at magnitude 1..2 is 2^-7
    else:
        got = A.whisper_log_mel_spectrogram(ctx, x, 128, dtype=m.F16).astype(np.float32)
        ulp = 2.0 ** -10
    np.testing.assert_allclose(got, ref, atol=ulp, rtol=0)


def test_s3_logmel_matches_oracle(ctx):
    from mlx_swift_audio_amd import audio as A
    x = _signals()["sweep"]
    got = A.s3_log_mel_spectrogram(ctx, x, 128)
    ref = O.s3_log_mel_spectrogram(x, 128)
    assert got.shape == ref.shape == (128, 200)
    np.testing.assert_allclose(got, ref, atol=TOL, rtol=0)


def test_error_paths(ctx):
    """Error behaviour mirrors the reference: too-short input is fatal in stft (S3TokenizerUtils.swift:248-251)."""
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import audio as A
    with pytest.raises(m.MiaError) as e:
        A.whisper_log_mel_spectrogram(ctx, np.zeros(100, np.float32), 80)
    assert e.value.code in (m._lib.ERR_INVALID_AUDIO,)
    with pytest.raises(m.MiaError):
        A.whisper_log_mel_spectrogram(ctx, O.synth_clip(0, 16000), 129)


def test_s3gen_mel_24k(ctx):
    """80-bin 24 kHz prompt-feature mel (S3GenMel.swift:43-102) against the oracle."""
    from mlx_swift_audio_amd import audio as A
    OL = O
    rng = np.random.default_rng(11)
    for n in (1200, 24000, 24000 * 6 + 77):
        t = np.arange(n, dtype=np.float32) / 24000.0
        y = (0.3 * np.sin(2 * np.pi * 220.0 * t) + 0.1 * np.sin(2 * np.pi * 3100.0 * t) + 0.05 * rng.standard_normal(n)).astype(np.float32)
        want = OL.s3gen_mel_spectrogram(y)
        got = A.s3gen_mel_spectrogram(ctx, y)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, atol=2e-3, rtol=0)      # natural-log units; 1920-term fp32 sums
    import mlx_swift_audio_amd as M
    with pytest.raises(M.MiaError):
        A.s3gen_mel_spectrogram(ctx, np.zeros(100, np.float32))


def test_resample_linear(ctx):
    """resampleAudio / linearInterpolate1d (CosyHiFTGenerator.swift:17-60) against the float32 oracle, bit for bit."""
    from mlx_swift_audio_amd import audio as A
    from oracle import hift as OH
    rng = np.random.default_rng(12)
    for n, (fr, to) in ((48000, (24000, 16000)), (16001, (16000, 24000)), (7, (24000, 16000)), (3, (48000, 16000))):
        x = rng.standard_normal(n).astype(np.float32)
        want = OH.linear_interpolate_1d(x[:, None], np.float32(to) / np.float32(fr))[:, 0]
        got = A.resample_audio(ctx, x, fr, to)
        assert got.shape == want.shape
        np.testing.assert_array_equal(got, want)
