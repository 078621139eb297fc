"""End-to-end CosyVoice2 chain on the GPU (SURVEY.md rows a15-a18 + the 24 kHz mel): reference clip -> S3 tokens + 80-mel prompt
features -> Qwen2LM RAS token generation -> flow (conformer + CFM) -> HiFT vocoder, through the host mirror of CosyVoice2Model.
Each hand-over is compared with the oracle of the stage that consumes it (the LM's sampled stream is checked in test_lm_gpu.py;
here its output feeds the oracles so that one sampling fork cannot mask a later stage)."""
import numpy as np
import pytest

from mlx_swift_audio_amd import synthetic as S

pytestmark = pytest.mark.gpu

S_TOK = 6561


def test_zero_shot_pipeline(ctx):
    import dataclasses
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import cosyvoice2 as CV, flow as HF, hift as HH, lm as HL, s3tok as HS, speaker as SP
    from oracle import campplus as OC, flow as OF, hift as OH, logmel as OL

    lcfg = S.LM_CONFIGS["qwen-micro"]
    lw = S.lm_weights(lcfg, seed=3, round_to="f16")
    lw.update(S.qwen2lm_extra_weights(lcfg, S_TOK, seed=3, round_to="f16"))
    llm = HL.Qwen2LM(HL.CausalLM.load(ctx, lcfg, lw, m.F16), lw, speech_token_size=S_TOK)
    fcfg = dataclasses.replace(S.FLOW_CONFIGS["flow_micro"], vocab_size=S_TOK)
    fw = S.flow_weights(fcfg, seed=3)
    flow = HF.FlowModule.load(ctx, fcfg, fw)
    hcfg = S.HIFT_CONFIGS["hift_micro"]
    hw = S.hift_weights(hcfg, seed=3)
    hift = HH.HiFTGenerator.load(ctx, hcfg, hw)
    scfg = S.S3_CONFIGS["s3_micro"]
    s3 = HS.S3Tokenizer.load(ctx, scfg, S.s3_weights(scfg, 3))
    cw = S.campplus_weights(3)
    spk_enc = SP.CAMPlusSpeakerEncoder.load(ctx, cw)
    model = CV.CosyVoice2Model(ctx, llm, flow, hift, s3, spk_enc)

    # ---- conditionals from a 2 s, 24 kHz reference clip (resampled to 16 kHz on the device for the tokenizer)
    ref24 = OL.synth_clip(2, 48000)
    rng = np.random.default_rng(0)
    cond = model.prepare_conditionals(ref24, prompt_text=[7, 8, 9])          # speaker embedding from the clip itself (CAM++)
    from mlx_swift_audio_amd import audio as A
    want_spk = OC.CAMPPlusOracle(cw).inference(A.resample_audio(ctx, ref24, 24000, 16000))[0]
    assert np.abs(cond.speaker_embedding - want_spk).max() <= 3e-3 * np.abs(want_spk).max()
    n_p = cond.prompt_speech_token.shape[0]
    assert n_p == 50 and cond.prompt_mel.shape == (2 * n_p, 80)          # 2 s -> 50 tokens @ 25 Hz, 100 mel frames @ 50 Hz
    np.testing.assert_allclose(cond.prompt_mel, OL.s3gen_mel_spectrogram(ref24).T[:2 * n_p], atol=2e-3)

    # ---- synthesize
    text = [21, 22, 23, 24, 25, 26, 27, 28]
    u = rng.random(4000).astype(np.float32)
    draws = {}

    def z_fn(T):
        draws["z"] = np.random.default_rng(1).standard_normal((80, T)).astype(np.float32)
        return draws["z"]

    def noise_fn(L):
        draws["noise"] = np.random.default_rng(2).standard_normal((L, 9)).astype(np.float32)
        return draws["noise"]

    audio, tokens = model.synthesize(text, cond, u, z_fn, noise_fn)
    assert len(text) * 2 - 1 <= len(tokens) <= len(text) * 20 and all(0 <= t < S_TOK for t in tokens)
    assert audio.shape == (2 * len(tokens) * 480,) and np.isfinite(audio).all() and np.abs(audio).max() <= 0.99 + 1e-7

    # ---- stage-wise parity on the same hand-overs
    mel = model.tokens_to_mel(np.asarray(tokens, np.int32), cond.prompt_speech_token, cond.prompt_mel, cond.speaker_embedding, draws["z"])
    want_mel, _ = OF.inference(fw, fcfg, np.asarray(tokens), cond.prompt_speech_token, cond.prompt_mel, cond.speaker_embedding, draws["z"])
    np.testing.assert_allclose(mel, want_mel, atol=2e-3, rtol=2e-3)
    want_audio, want_src = OH.vocode(hw, hcfg, mel, draws["noise"])
    np.testing.assert_allclose(audio, want_audio, atol=3e-2)                # f0 rounding moves the source phase (see test_hift_gpu.py)
    pinned, _ = hift(mel, cache_source=want_src, noise=draws["noise"])
    np.testing.assert_allclose(pinned, want_audio, atol=3e-4, rtol=1e-3)
    again, tokens2 = model.synthesize(text, cond, u, z_fn, noise_fn)
    assert tokens2 == tokens
    np.testing.assert_array_equal(again, audio)
    # ---- the other three modes are the same stages with different prompts (CosyVoice2Model.swift:253-397)
    xa, xt = model.synthesize_cross_lingual(text, cond, u, z_fn, noise_fn)
    assert xt == llm.inference(text, [], [], u) and xa.shape == (2 * len(xt) * 480,) and np.isfinite(xa).all()
    ia, it = model.synthesize_instruct(text, [3, 4], cond, u, z_fn, noise_fn)
    assert it == llm.inference(text, [3, 4], [], u) and ia.shape == (2 * len(it) * 480,)
    src = model.tokenize_speech(OL.synth_clip(9, 36000))                      # 1.5 s source clip -> 37-38 tokens
    assert 36 <= src.shape[0] <= 38 and (0 <= src).all() and (src < S_TOK).all()
    va = model.synthesize_vc(src, cond, z_fn, noise_fn)
    want_vc, _ = OF.inference(fw, fcfg, src, cond.prompt_speech_token, cond.prompt_mel, cond.speaker_embedding, draws["z"])
    got_vc = model.tokens_to_mel(src, cond.prompt_speech_token, cond.prompt_mel, cond.speaker_embedding, draws["z"])
    np.testing.assert_allclose(got_vc, want_vc, atol=2e-3, rtol=2e-3)
    assert va.shape == (2 * src.shape[0] * 480,) and np.abs(va).max() <= 0.99 + 1e-7
    # ---- several sentences: one pass of the flow for all of them (synthesize_batch), each equal to its own synthesize()
    texts = [text, [30, 31, 32], [40, 41, 42, 43, 44]]
    us = [u, rng.random(4000).astype(np.float32), rng.random(4000).astype(np.float32)]
    z_of = lambda T: np.random.default_rng(1).standard_normal((80, T)).astype(np.float32)          # same draw rule as z_fn
    n_of = lambda L: np.random.default_rng(2).standard_normal((L, 9)).astype(np.float32)
    many = model.synthesize_batch(texts, cond, us, z_of, n_of)
    for (ba, bt), t_, u_ in zip(many, texts, us):
        sa, st_ = model.synthesize(t_, cond, u_, z_of, n_of)
        assert bt == st_
        np.testing.assert_array_equal(ba, sa)
    for h in (flow, hift, s3, spk_enc):
        h.close()
    llm.lm.close()
