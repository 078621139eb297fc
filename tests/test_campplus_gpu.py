"""GPU parity: CAM++ speaker encoder + Kaldi filterbank (HIP, through the C ABI) vs the fp32 CPU oracle on seeded random-init
weights of the real architecture (7 M parameters, 12/24/16-layer dense blocks).

Tolerances: fbank |delta| <= 2e-3 in the log domain on bins that carry signal (the oracle's FFT is float64, the device's DFT is an
exact-fp32 MFMA contraction; empty low triangles are exactly log(FLT_EPSILON) on both sides); embedding |delta| <= 2e-3 of its
own scale through 60+ fp32 layers (folded BatchNorm, different summation order)."""
import numpy as np
import pytest

from mlx_swift_audio_amd import synthetic as S
from oracle import campplus as OC
from oracle import logmel as OL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def enc(ctx):
    from mlx_swift_audio_amd import speaker as SP
    w = S.campplus_weights(7)
    e = SP.CAMPlusSpeakerEncoder.load(ctx, {"campplus." + k: v for k, v in w.items()})   # prefixed keys, as in the checkpoint
    yield e, OC.CAMPPlusOracle(w)
    e.close()


@pytest.mark.parametrize("n", [400, 559, 16000, 40000])
def test_kaldi_fbank_matches_oracle(enc, n):
    e, _ = enc
    x = OL.synth_clip(n % 7, 48000)[:n]
    got = e.extract_fbank(x)
    want = OC.kaldi_fbank(x)
    assert got.shape == want.shape == ((n - 400) // 160 + 1, 80)
    np.testing.assert_allclose(got, want, atol=2e-3)
    gm = e.extract_fbank(x, mean_norm=True)
    np.testing.assert_allclose(gm, want - want.mean(axis=0, keepdims=True), atol=2e-3)


@pytest.mark.parametrize("T", [1, 7, 98, 250, 601])
def test_forward_matches_oracle(enc, T):
    e, ora = enc
    feats = (np.random.default_rng(T).standard_normal((T, 80)) * 2.0).astype(np.float32)
    got = e.forward(feats)
    want = ora.forward(feats[None])[0]
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-3 * scale, (np.abs(got - want).max(), scale)


def test_embed_end_to_end_and_determinism(enc):
    e, ora = enc
    x = OL.synth_clip(4, 6 * 16000)
    got = e(x)
    want = ora.inference(x)
    assert got.shape == (1, 192)
    assert np.abs(got - want).max() <= 3e-3 * np.abs(want).max()
    np.testing.assert_array_equal(e(x), got)
    assert np.abs(e(OL.synth_clip(5, 6 * 16000)) - got).max() > 1e-3 * np.abs(want).max()


def test_unloaded_encoder_returns_zeros_and_bad_input_fails_loudly(ctx, enc):
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import speaker as SP
    empty = SP.CAMPlusSpeakerEncoder.load(ctx, {})
    assert not empty.is_loaded and np.array_equal(empty(np.zeros(16000, np.float32)), np.zeros((1, 192), np.float32))
    e, _ = enc
    with pytest.raises(m.MiaError):
        e(np.zeros(399, np.float32))                                   # shorter than one frame
    w = S.campplus_weights(7)
    del w["blocks.2.layers.15.cam_layer.linear2.bias"]
    with pytest.raises(m.MiaError, match="linear2.bias"):
        SP.CAMPlusSpeakerEncoder.load(ctx, w)
