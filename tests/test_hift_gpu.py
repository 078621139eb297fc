"""GPU parity of the HiFT vocoder (row a17) against oracle/hift.py, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import mlx_swift_audio_amd as M
    from mlx_swift_audio_amd import hift as HH, synthetic as S
    ctx = M.Context()
    cfg = S.HIFT_CONFIGS["hift_micro"]
    w = S.hift_weights(cfg)
    gen = HH.HiFTGenerator.load(ctx, cfg, w)
    return ctx, cfg, w, gen


def _mel(T, seed=0):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((80, T)) * 1.5 - 2).astype(np.float32)


def test_f0_predictor(env):
    from oracle import hift as OH
    ctx, cfg, w, gen = env
    for T in (2, 7, 50):
        mel = _mel(T, T)
        got, want = gen.f0_predictor(mel), OH.f0_predictor(w, mel)
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-3)      # fp32 MFMA vs torch fp32 summation order; f0 ~ 1e2


def test_source_module(env):
    from oracle import hift as OH
    ctx, cfg, w, gen = env
    rng = np.random.default_rng(5)
    for T in (2, 33, 400):
        f0 = np.abs(rng.standard_normal(T) * 110).astype(np.float32)
        f0[rng.random(T) < 0.2] = 3.0                     # unvoiced frames
        noise = rng.standard_normal((T * 480, 9)).astype(np.float32)
        want = OH.source(w, cfg, f0, noise)
        got = gen.m_source(f0, noise)
        # the float32 operation order of the phase track is replayed exactly; what is left is sin / tanh ulp differences
        np.testing.assert_allclose(got, want, atol=5e-6)
        got0, want0 = gen.m_source(f0, None), OH.source(w, cfg, f0, None)
        np.testing.assert_allclose(got0, want0, atol=5e-6)


def test_decode(env):
    from oracle import hift as OH
    ctx, cfg, w, gen = env
    rng = np.random.default_rng(6)
    for T in (2, 5, 40):
        mel = _mel(T, 100 + T)
        s = np.tanh(rng.standard_normal(T * 480)).astype(np.float32) * 0.3
        want = OH.decode(w, cfg, mel, s)
        got = gen.decode(mel, s)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, atol=3e-4, rtol=1e-3)     # ~80 fp32 convolutions deep, exp() at the end


def test_vocode_end_to_end_and_cache(env):
    from oracle import hift as OH
    ctx, cfg, w, gen = env
    rng = np.random.default_rng(7)
    T = 12
    mel = _mel(T, 9)
    noise = rng.standard_normal((T * 480, 9)).astype(np.float32)
    want_pcm, want_s = OH.vocode(w, cfg, mel, noise)
    pcm, s = gen(mel, noise=noise)
    # f0 differs by fp32 rounding and is multiplied by up to 9 * 2 pi * 480 / 24000 per frame into the phase: short clip, looser bound
    np.testing.assert_allclose(s, want_s, atol=2e-3)
    np.testing.assert_allclose(pcm, want_pcm, atol=2e-2)
    cache = want_s[:1000].copy()
    pcm2, s2 = gen(mel, cache_source=cache, noise=noise)
    np.testing.assert_array_equal(s2[:1000], cache)
    np.testing.assert_array_equal(s2[1000:], s[1000:])
    # a full-length cache pins the source completely: decode parity at the tight tolerance
    pcm3, s3 = gen(mel, cache_source=want_s, noise=noise)
    np.testing.assert_array_equal(s3, want_s)
    np.testing.assert_allclose(pcm3, want_pcm, atol=3e-4, rtol=1e-3)
    pcm4, _ = gen(mel, cache_source=want_s, noise=noise)
    np.testing.assert_array_equal(pcm3, pcm4)          # deterministic (gather overlap-add, no atomics)


def test_full_size_config(env):
    from oracle import hift as OH
    from mlx_swift_audio_amd import hift as HH, synthetic as S
    ctx = env[0]
    cfg = S.HIFT_CONFIGS["hift_cosyvoice2"]
    w = S.hift_weights(cfg, seed=1)
    gen = HH.HiFTGenerator.load(ctx, cfg, w)
    rng = np.random.default_rng(8)
    T = 60
    mel = _mel(T, 21)
    s = np.tanh(rng.standard_normal(T * 480)).astype(np.float32) * 0.3
    got, want = gen.decode(mel, s), OH.decode(w, cfg, mel, s)
    np.testing.assert_allclose(got, want, atol=3e-4, rtol=1e-3)
    assert np.abs(got).max() <= 0.99 + 1e-7
    # long clip: bounded, finite, deterministic
    T = 1500
    mel = _mel(T, 22)
    a, sa = gen(mel)
    b, sb = gen(mel)
    assert a.shape == (T * 480,) and np.isfinite(a).all() and np.abs(a).max() <= 0.99 + 1e-7
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(sa, sb)
    # production configuration, stacked: three utterances of different lengths in one pass = their single calls, bit for bit
    _check_batch(gen, [250, 97, 400], 31)
    gen.close()


def _check_batch(gen, Ts, seed, with_noise=True):
    rng = np.random.default_rng(seed)
    mels = [_mel(T, seed + 7 * i) for i, T in enumerate(Ts)]
    noises = [rng.standard_normal((T * gen.up, gen.cfg.nb_harmonics + 1)).astype(np.float32) for T in Ts] if with_noise else None
    got = gen.vocode_batch(mels, noises)
    assert len(got) == len(Ts)
    for u, T in enumerate(Ts):
        want, _ = gen(mels[u], noise=noises[u] if with_noise else None)
        assert got[u].shape == (T * gen.up,)
        assert np.array_equal(got[u], want), (u, T, np.abs(got[u] - want).max())


def test_vocode_batch_equals_single_calls(env):
    """mia_hift_vocode_batch: utterances of different lengths stacked into one pass (every convolution once over grid.z = sequence x
    phase, rows past a sequence's end read as zero) give, utterance by utterance, the bits of their own single calls -- shortest and
    longest first, a length-2 utterance next to long ones, with and without the additive noise."""
    ctx, cfg, w, gen = env
    _check_batch(gen, [40, 17, 40, 2, 63], 11)
    _check_batch(gen, [5, 90], 12, with_noise=False)
    _check_batch(gen, [33], 13)                          # n = 1 goes through the same entry point
    # and against the oracle directly
    from oracle import hift as OH
    rng = np.random.default_rng(14)
    mels = [_mel(21, 1), _mel(48, 2)]
    noises = [rng.standard_normal((T * 480, 9)).astype(np.float32) for T in (21, 48)]
    got = gen.vocode_batch(mels, noises)
    for u in range(2):
        src = gen(mels[u], noise=noises[u])[1]            # (f0's last-ulp error is amplified in the phase track: pin the source)
        np.testing.assert_allclose(got[u], OH.decode(w, cfg, mels[u], src), atol=3e-4, rtol=1e-3)


def test_errors(env):
    import mlx_swift_audio_amd as M
    from mlx_swift_audio_amd import hift as HH
    ctx, cfg, w, gen = env
    with pytest.raises(M.MiaError):
        gen.f0_predictor(_mel(1))
    with pytest.raises(M.MiaError):
        gen.decode(_mel(4), np.zeros(5, np.float32))
    bad = dict(w); bad.pop("conv_post.weight")
    with pytest.raises(M.MiaError):
        HH.HiFTGenerator.load(ctx, cfg, bad)
