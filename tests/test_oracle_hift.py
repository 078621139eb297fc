"""CPU checks that pin the HiFT oracle's building blocks against independent implementations (torch.stft / torch.istft,
torch.nn.functional.interpolate, an explicit-loop transposed convolution).  The reference ships no golden vectors for this path."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import hift as OH
from mlx_swift_audio_amd import synthetic as S


def test_linear_interpolate_matches_torch():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((37, 3)).astype(np.float32)
    for scale in (4.0, 480.0):
        got = OH.linear_interpolate_1d(x, np.float32(scale))
        want = F.interpolate(torch.from_numpy(x.T)[None], scale_factor=scale, mode="linear", align_corners=False)[0].T.numpy()
        # the Swift clips the source index at T - 1.001 (not T - 1): the last half-cell differs by at most 1e-3 of a step
        np.testing.assert_allclose(got, want, atol=2e-3 * np.abs(np.diff(x, axis=0)).max())
    down = OH.linear_interpolate_1d(np.repeat(x, 480, axis=0), np.float32(1.0) / np.float32(480))
    np.testing.assert_array_equal(down, x)          # piecewise-constant track: the 480:1 downsample returns the frame values


def test_random_initial_phase_is_inert():
    cfg = S.HIFT_CONFIGS["hift_micro"]
    rng = np.random.default_rng(1)
    f0 = np.abs(rng.standard_normal(12) * 120).astype(np.float32)
    f0_up = np.repeat(f0, cfg.upsample_factor)
    a = OH.sine_gen2(f0_up, cfg, None, None)
    b = OH.sine_gen2(f0_up, cfg, rng.random(9).astype(np.float32), None)
    np.testing.assert_array_equal(a, b)


def test_stft_istft_against_torch():
    rng = np.random.default_rng(2)
    x = rng.standard_normal(4 * 60).astype(np.float32)
    re, im = OH.stft(x)
    win = torch.from_numpy(OH.hann_periodic(16))
    ref = torch.stft(torch.from_numpy(x), 16, 4, 16, win, center=True, pad_mode="reflect", return_complex=True)
    np.testing.assert_allclose(re, ref.real.numpy(), atol=2e-5)
    np.testing.assert_allclose(im, ref.imag.numpy(), atol=2e-5)
    mag, ph = np.abs(re + 1j * im).astype(np.float32), np.angle(re + 1j * im).astype(np.float32)
    back = OH.istft(mag, ph)
    np.testing.assert_allclose(back, x, atol=2e-5)
    want = torch.istft(ref, 16, 4, 16, win, center=True, length=x.shape[0]).numpy()
    np.testing.assert_allclose(back, want, atol=2e-5)


def test_transposed_conv_definition():
    """MLX convTransposed1d semantics restated as an explicit scatter: y[t s + k - p, co] += x[t, ci] w[co, k, ci]."""
    rng = np.random.default_rng(3)
    for (K, s) in ((16, 8), (11, 5), (7, 3)):
        p = (K - s) // 2
        x = rng.standard_normal((9, 4)).astype(np.float32)
        w = rng.standard_normal((5, K, 4)).astype(np.float32)
        T_out = (x.shape[0] - 1) * s - 2 * p + K
        y = np.zeros((T_out, 5), np.float64)
        for t in range(x.shape[0]):
            for k in range(K):
                o = t * s + k - p
                if 0 <= o < T_out:
                    y[o] += w[:, k, :].astype(np.float64) @ x[t].astype(np.float64)
        got = F.conv_transpose1d(torch.from_numpy(x.T)[None], torch.from_numpy(w).permute(2, 0, 1), stride=s, padding=p)[0].T.numpy()
        np.testing.assert_allclose(got, y, atol=1e-4)


def test_vocode_shapes_and_cache():
    cfg = S.HIFT_CONFIGS["hift_micro"]
    w = S.hift_weights(cfg)
    rng = np.random.default_rng(4)
    T = 6
    mel = (rng.standard_normal((80, T)) * 1.5 - 2).astype(np.float32)
    noise = rng.standard_normal((T * 480, 9)).astype(np.float32)
    pcm, s = OH.vocode(w, cfg, mel, noise)
    assert pcm.shape == (T * 480,) and s.shape == (T * 480,)
    assert np.abs(pcm).max() <= 0.99 + 1e-7
    cache = rng.standard_normal(700).astype(np.float32) * 0.1
    pcm2, s2 = OH.vocode(w, cfg, mel, noise, cache_source=cache)
    np.testing.assert_array_equal(s2[:700], cache)
    np.testing.assert_array_equal(s2[700:], s[700:])
