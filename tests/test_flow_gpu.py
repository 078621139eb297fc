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


@pytest.mark.parametrize("finalize,enc_chunk,dec_chunk", [(False, 0, 0), (True, 25, 50), (False, 7, 10), (True, 1, 1), (True, 64, 37)])
def test_streaming_masks_and_finalize(env, finalize, enc_chunk, dec_chunk):
    """The modules' chunked-synthesis switches (mia_flow_inference_streaming): finalize = false trims the encoder look-ahead, the static
    chunk sizes turn on the block-causal attention masks of the conformer encoder (chunk, chunk * 2 after up-sampling) and of the
    estimator's transformer blocks.  Chunk sizes that do not divide the 32-query tiles, chunk 1 (pure causal) and the checkpoint's
    25 / 50 are covered; with masks on, later tokens must not influence earlier chunks."""
    from oracle import flow as OF
    ctx, cfg, w, mod = env
    n, m, m1 = 61, 14, 28
    tok, ptok, pf, emb, _ = _inputs(cfg, n, m, m1, 9)
    T = 2 * (n + m) - (0 if finalize else cfg.pre_lookahead_len * 2)
    z = np.random.default_rng(10).standard_normal((80, T)).astype(np.float32)
    want, _ = OF.inference(w, cfg, tok, ptok, pf, emb, z, n_timesteps=3, finalize=finalize, enc_static_chunk=enc_chunk, dec_static_chunk=dec_chunk)
    got = mod.inference_streaming(tok, ptok, pf, emb, z, n_timesteps=3, finalize=finalize, enc_static_chunk=enc_chunk, dec_static_chunk=dec_chunk)
    assert got.shape == want.shape == (80, T - m1)
    np.testing.assert_allclose(got, want, atol=1e-3, rtol=2e-3)
    if finalize and enc_chunk == 0 and dec_chunk == 0:
        np.testing.assert_array_equal(got, mod.inference(tok, ptok, pf, emb, z, n_timesteps=3))


def test_attention_chunk_mask_kernel(env):
    """mia_op_attention_f32 has no mask argument; the masked kernel is exercised through the flow above.  Here: the streaming flow with huge
    chunks equals the unmasked flow bit for bit (the mask path itself changes nothing when no key is excluded)."""
    ctx, cfg, w, mod = env
    tok, ptok, pf, emb, z = _inputs(cfg, 40, 10, 20, 12)
    a = mod.inference(tok, ptok, pf, emb, z, n_timesteps=2)
    b = mod.inference_streaming(tok, ptok, pf, emb, z, n_timesteps=2, finalize=True, enc_static_chunk=4096, dec_static_chunk=4096)
    np.testing.assert_array_equal(a, b)


def _assert_batch_equals_single(mod, utts, n_timesteps):
    singles = [mod.inference(*u, n_timesteps=n_timesteps) for u in utts]
    batch = mod.inference_batch(utts, n_timesteps=n_timesteps)
    assert len(batch) == len(singles)
    for i, (got, want) in enumerate(zip(batch, singles)):
        assert got.shape == want.shape
        assert np.array_equal(got, want), (i, float(np.abs(got - want).max()))       # bit for bit: padding never reaches a valid frame
    # order and company do not matter either
    rev = mod.inference_batch(utts[::-1], n_timesteps=n_timesteps)
    for got, want in zip(rev[::-1], singles):
        assert np.array_equal(got, want)
    return singles


def test_inference_batch_equals_single_calls(env):
    """mia_flow_inference_batch: utterances of different lengths, prompt sizes and speakers stacked into one pass give, utterance by
    utterance, the bits of their own mia_flow_inference call (which the tests above check against the oracle): the look-ahead window,
    the attention keys and every convolution stay inside the utterance."""
    from oracle import flow as OF
    ctx, cfg, w, mod = env
    utts = [_inputs(cfg, n, m, m1, seed) for (n, m, m1, seed) in ((30, 12, 24, 2), (5, 0, 0, 3), (90, 40, 80, 4), (1, 0, 0, 5), (64, 3, 6, 6))]
    singles = _assert_batch_equals_single(mod, utts, 3)
    want, _ = OF.inference(w, cfg, *utts[0], n_timesteps=3)                           # and the batch is the oracle's answer too
    np.testing.assert_allclose(singles[0], want, atol=1e-3, rtol=1e-3)
    one = mod.inference_batch(utts[2:3], n_timesteps=3)                               # a batch of one is the single call
    assert np.array_equal(one[0], singles[2])
    import mlx_swift_audio_amd as M
    with pytest.raises(M.MiaError):
        mod.inference_batch([utts[0]] * 65, n_timesteps=1)


def test_inference_batch_production_config():
    """The same identity at CosyVoice2's production flow configuration (512-wide conformer, 256-channel estimator), 2 Euler steps."""
    import mlx_swift_audio_amd as M
    from mlx_swift_audio_amd import flow as HF, synthetic as S
    ctx = M.Context()
    cfg = S.FLOW_CONFIGS["flow_cosyvoice2"]
    mod = HF.FlowModule.load(ctx, cfg, S.flow_weights(cfg))
    utts = [_inputs(cfg, n, m, m1, seed) for (n, m, m1, seed) in ((120, 50, 100, 11), (37, 50, 100, 12), (200, 0, 0, 13))]
    _assert_batch_equals_single(mod, utts, 2)
    mod.close()
