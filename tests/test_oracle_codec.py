"""CPU tests pinning the codec oracle (oracle/codec.py): shapes stated in SURVEY.md section 8(a) rows a13/a14 and the
transposed-convolution restatement against an explicit scatter-add definition."""
import numpy as np
import torch

from mlx_swift_audio_amd import synthetic as S
from oracle import codec as OC


def test_convt_matches_scatter_definition():
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((1, 3, 5)).astype(np.float32))
    w = torch.from_numpy(rng.standard_normal((4, 6, 3)).astype(np.float32))       # MLX layout [Cout,K,Cin]
    s, p = 3, 2
    y = OC._convt1d_cf(x, w, None, s, p)[0].numpy()
    T_out = (5 - 1) * s - 2 * p + 6
    ref = np.zeros((4, T_out), np.float32)
    for t in range(5):
        for k in range(6):
            j = t * s + k - p
            if 0 <= j < T_out:
                ref[:, j] += w[:, k, :].numpy() @ x[0, :, t].numpy()
    np.testing.assert_allclose(y, ref, atol=1e-5)


def test_snac_shapes_and_noise_len():
    cfg = S.SNAC_CONFIGS["snac_micro"]
    o = OC.SNACOracle(cfg, S.snac_weights(cfg, 1))
    codes = [[1, 2, 3], [4, 5, 6, 7, 8, 9]]                      # n, 2n with vq strides (2, 1)
    pcm = o.decode(codes)
    hop = int(np.prod(cfg.decoder_rates))
    assert pcm.shape == (6 * hop,) and np.all(np.abs(pcm) <= 1.0)
    nl = o.noise_len(6)
    assert nl == 6 * 4 + 6 * 8
    rng = np.random.default_rng(2)
    pcm2 = o.decode(codes, rng.standard_normal(nl).astype(np.float32))
    assert np.abs(pcm2 - pcm).max() > 1e-4                       # the noise path is live
    # a level with the wrong expanded length is skipped, like embedCodes (SNACDecoder.swift:397-401)
    z = o.embed_codes([[1, 2], [4, 5, 6, 7, 8, 9]])
    z1 = o.embed_codes([[], [4, 5, 6, 7, 8, 9]])
    assert torch.equal(z, z1)


def test_real_snac_geometry():
    # a13: 1200 tokens -> N = 171 frames -> codes (N, 2N, 4N) -> 2048*N samples
    cfg = S.SNAC_CONFIGS["snac_24khz"]
    T = 171 * 4
    for s in cfg.decoder_rates:
        T = (T - 1) * s - 2 * int(np.ceil(s / 2)) + 2 * s
    assert T == 171 * 2048 == 350208


def test_dac_shapes():
    cfg = S.DAC_CONFIGS["dac_micro"]
    o = OC.DACOracle(cfg, S.dac_weights(cfg, 1))
    codes = np.random.default_rng(0).integers(0, cfg.codebook_size, (2, 7))
    pcm = o.decode_from_codes(codes)
    # odd stride 5 with padding 3 and no output_padding loses one sample per input step boundary: (T*4)*5 - 1
    assert pcm.shape == ((7 * 4) * 5 - 1,) and np.all(np.abs(pcm) <= 1.0)
