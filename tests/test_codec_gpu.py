"""GPU parity: SNAC / DAC decoders (HIP, through the C ABI) vs the fp32 CPU oracle on the same seeded random-init weights,
codes and explicit noise.  Both sides are fp32; the tolerance (max |delta| <= 2e-4 on tanh-bounded samples in [-1,1]) covers
accumulation order and sinf/tanhf implementation differences."""
import numpy as np
import pytest

from mlx_swift_audio_amd import synthetic as S
from oracle import codec as OC

pytestmark = pytest.mark.gpu
TOL = 2e-4


@pytest.mark.parametrize("name", ["snac_micro", "snac_micro_cn"])
@pytest.mark.parametrize("with_noise", [False, True])
def test_snac_decode_matches_oracle(ctx, name, with_noise):
    from mlx_swift_audio_amd import codec as HC
    cfg = S.SNAC_CONFIGS[name]
    w = S.snac_weights(cfg, seed=3)
    dec = HC.SNACDecoder.load(ctx, cfg, w)
    ora = OC.SNACOracle(cfg, w)
    rng = np.random.default_rng(5)
    n = 37                                                        # ragged vs the 128-row GEMM tile on purpose
    codes = [rng.integers(0, cfg.codebook_size, n * (cfg.vq_strides[0] // s)).tolist() for s in cfg.vq_strides]
    T0 = n * cfg.vq_strides[0]
    assert dec.noise_len(T0) == ora.noise_len(T0)
    noise = rng.standard_normal(dec.noise_len(T0)).astype(np.float32) if with_noise else None
    got = dec.decode(codes, noise)
    ref = ora.decode(codes, noise)
    assert got.shape == ref.shape == (T0 * int(np.prod(cfg.decoder_rates)),)
    assert np.abs(got - ref).max() <= TOL, np.abs(got - ref).max()
    dec.close()


def test_snac_level_with_wrong_length_is_skipped(ctx):
    from mlx_swift_audio_amd import codec as HC
    cfg = S.SNAC_CONFIGS["snac_micro"]
    w = S.snac_weights(cfg, seed=3)
    dec = HC.SNACDecoder.load(ctx, cfg, w)
    ora = OC.SNACOracle(cfg, w)
    codes = [[1, 2], [4, 5, 6, 7, 8, 9]]
    np.testing.assert_allclose(dec.decode(codes), ora.decode(codes), atol=TOL)
    dec.close()


def test_dac_decode_matches_oracle(ctx):
    from mlx_swift_audio_amd import codec as HC
    cfg = S.DAC_CONFIGS["dac_micro"]
    w = S.dac_weights(cfg, seed=4)
    dec = HC.DACCodec.load(ctx, cfg, w)
    ora = OC.DACOracle(cfg, w)
    codes = np.random.default_rng(1).integers(0, cfg.codebook_size, (2, cfg.n_codebooks, 45))
    got = dec.decode_from_codes(codes)
    for b in range(2):
        ref = ora.decode_from_codes(codes[b])
        assert got[b].shape == ref.shape
        assert np.abs(got[b] - ref).max() <= TOL, np.abs(got[b] - ref).max()
    dec.close()


def test_snac_24khz_real_geometry(ctx):
    """mlx-community/snac_24khz shapes (768-d latent, 1024 -> 64 channels, rates 8.8.4.2, depthwise): ~1 s of audio vs the oracle.
    The tolerance is the micro-model one scaled by the depth of the 1024-channel stack (fp32 both sides, different sum orders)."""
    from mlx_swift_audio_amd import codec as HC
    cfg = S.SNAC_CONFIGS["snac_24khz"]
    w = S.snac_weights(cfg, seed=6)
    dec = HC.SNACDecoder.load(ctx, cfg, w)
    ora = OC.SNACOracle(cfg, w)
    rng = np.random.default_rng(8)
    n = 12                                                        # 12 frames -> 24 576 samples
    codes = [rng.integers(0, cfg.codebook_size, n * (cfg.vq_strides[0] // s)).tolist() for s in cfg.vq_strides]
    T0 = n * cfg.vq_strides[0]
    noise = rng.standard_normal(dec.noise_len(T0)).astype(np.float32)
    got, ref = dec.decode(codes, noise), ora.decode(codes, noise)
    assert got.shape == ref.shape == (T0 * int(np.prod(cfg.decoder_rates)),)
    assert np.abs(got - ref).max() <= 5e-4, np.abs(got - ref).max()
    dec.close()


def test_dac_speech_real_geometry(ctx):
    """The DAC speech configuration (1536 -> 96 channels, rates 8.5.4.2, 2 x 1024 codebooks): 60 code steps vs the oracle."""
    from mlx_swift_audio_amd import codec as HC
    cfg = S.DAC_CONFIGS["dac_speech"]
    w = S.dac_weights(cfg, seed=7)
    dec = HC.DACCodec.load(ctx, cfg, w)
    ora = OC.DACOracle(cfg, w)
    codes = np.random.default_rng(2).integers(0, cfg.codebook_size, (1, cfg.n_codebooks, 60))
    got = dec.decode_from_codes(codes)
    ref = ora.decode_from_codes(codes[0])
    assert got[0].shape == ref.shape
    assert np.abs(got[0] - ref).max() <= 5e-4, np.abs(got[0] - ref).max()
    dec.close()


def test_codec_error_paths(ctx):
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import codec as HC
    cfg = S.SNAC_CONFIGS["snac_micro"]
    w = S.snac_weights(cfg, seed=3)
    bad = dict(w)
    del bad["decoder.model.layers.1.bias"]
    with pytest.raises(m.MiaError):
        HC.SNACDecoder.load(ctx, cfg, bad)
    dec = HC.SNACDecoder.load(ctx, cfg, w)
    with pytest.raises(m.MiaError):
        dec.decode([[cfg.codebook_size + 5], [1, 2]])            # code out of range
    with pytest.raises(m.MiaError):
        dec.decode([[1], [1, 2]], np.zeros(3, np.float32))       # wrong noise length
    dec.close()
