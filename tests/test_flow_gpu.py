"""GPU parity of the CosyVoice2 flow (row a16) and of the fp32 flash attention against oracle/flow.py, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import mlx_swift_audio_amd as M
    from mlx_swift_audio_amd import flow as HF, synthetic as S
    ctx = M.Context()
    cfg = S.FLOW_CONFIGS["flow_micro"]
    w = S.flow_weights(cfg)
    mod = HF.FlowModule.load(ctx, cfg, w)
    return ctx, cfg, w, mod


def _inputs(cfg, n, m, m1, seed):
    rng = np.random.default_rng(seed)
    tok = rng.integers(0, cfg.vocab_size, n).astype(np.int32)
    ptok = rng.integers(0, cfg.vocab_size, m).astype(np.int32)
    pf = rng.standard_normal((m1, 80)).astype(np.float32)
    emb = rng.standard_normal(cfg.spk_embed_dim).astype(np.float32)
    z = rng.standard_normal((80, 2 * (n + m))).astype(np.float32)
    return tok, ptok, pf, emb, z


def test_encoder(env):
    import torch
    from oracle import flow as OF
    ctx, cfg, w, mod = env
    for n in (3, 17, 70, 150):
        tok = np.random.default_rng(n).integers(-2, cfg.vocab_size + 3, n).astype(np.int32)     # out-of-range ids are clipped
        ids = np.clip(tok, 0, cfg.vocab_size - 1).astype(np.int64)
        enc = OF.encoder(w, cfg, OF._t(w["input_embedding.weight"])[torch.from_numpy(ids)])
        want = OF._lin(w, "encoder_proj", enc).numpy()
        got = mod.encode(tok)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, atol=3e-4, rtol=1e-3)


def test_single_euler_step(env):
    """n_timesteps = 1: x1 = z + dt * cfg(estimator(z)) -- the estimator in isolation."""
    from oracle import flow as OF
    ctx, cfg, w, mod = env
    tok, ptok, pf, emb, z = _inputs(cfg, 21, 9, 18, 1)
    want, _ = OF.inference(w, cfg, tok, ptok, pf, emb, z, n_timesteps=1)
    got = mod.inference(tok, ptok, pf, emb, z, n_timesteps=1)
    np.testing.assert_allclose(got, want, atol=5e-4, rtol=1e-3)


def test_inference(env):
    from oracle import flow as OF
    ctx, cfg, w, mod = env
    for (n, m, m1, seed) in ((30, 12, 24, 2), (5, 0, 0, 3), (90, 40, 80, 4)):
        tok, ptok, pf, emb, z = _inputs(cfg, n, m, m1, seed)
        want, _ = OF.inference(w, cfg, tok, ptok, pf, emb, z)
        got = mod.inference(tok, ptok, pf, emb, z)
        assert got.shape == want.shape == (80, 2 * (n + m) - m1)
        np.testing.assert_allclose(got, want, atol=2e-3, rtol=2e-3)      # 10 Euler steps through a 60-GEMM-deep fp32 network
    again = mod.inference(tok, ptok, pf, emb, z)
    np.testing.assert_array_equal(got, again)


def test_full_size_config(env):
    from oracle import flow as OF
    from mlx_swift_audio_amd import flow as HF, synthetic as S
    ctx = env[0]
    cfg = S.FLOW_CONFIGS["flow_cosyvoice2"]
    w = S.flow_weights(cfg, seed=1)
    mod = HF.FlowModule.load(ctx, cfg, w)
    tok, ptok, pf, emb, z = _inputs(cfg, 40, 25, 50, 5)
    want, _ = OF.inference(w, cfg, tok, ptok, pf, emb, z, n_timesteps=2)
    got = mod.inference(tok, ptok, pf, emb, z, n_timesteps=2)
    np.testing.assert_allclose(got, want, atol=2e-3, rtol=2e-3)
    # long utterance at the default 10 steps: finite, deterministic
    tok, ptok, pf, emb, z = _inputs(cfg, 600, 150, 300, 6)
    a = mod.inference(tok, ptok, pf, emb, z)
    b = mod.inference(tok, ptok, pf, emb, z)
    assert a.shape == (80, 1200) and np.isfinite(a).all()
    np.testing.assert_array_equal(a, b)
    mod.close()


def test_errors(env):
    import mlx_swift_audio_amd as M
    from mlx_swift_audio_amd import flow as HF
    ctx, cfg, w, mod = env
    tok, ptok, pf, emb, z = _inputs(cfg, 4, 2, 4, 7)
    with pytest.raises(M.MiaError):
        mod.inference(tok, ptok, pf, emb, z[:, :-1])
    with pytest.raises(M.MiaError):
        mod.inference(tok, ptok, np.zeros((12, 80), np.float32), emb, z)       # prompt_feat as long as the whole sequence
    bad = dict(w); bad.pop("encoder_proj.weight")
    with pytest.raises(M.MiaError):
        HF.FlowModule.load(ctx, cfg, bad)
