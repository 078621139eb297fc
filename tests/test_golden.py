"""Frozen vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU restatement -- NOT reference outputs, see that
script's header).  CPU half: the oracle still reproduces them (drift guard).  GPU half: the HIP path reproduces them through the
C ABI at the stage tolerances of DESIGN.md section 4."""
import os

import numpy as np
import pytest

from mlx_swift_audio_amd import synthetic as S

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def _wsum(w, keys):
    return np.asarray([float(np.asarray(w[k], np.float64).sum()) for k in keys], np.float64)


# ---- CPU: oracle vs frozen -------------------------------------------------------------------------------------------------------
def test_oracle_logmel_golden():
    from oracle import logmel as OL
    g = _load("logmel.npz")
    np.testing.assert_array_equal(OL.synth_clip(3, 16000), g["clip"])
    np.testing.assert_allclose(OL.whisper_log_mel_spectrogram(g["clip"], 80), g["whisper80"], atol=1e-5)
    np.testing.assert_allclose(OL.whisper_log_mel_spectrogram(g["clip"], 128), g["whisper128"], atol=1e-5)
    np.testing.assert_allclose(OL.s3_log_mel_spectrogram(g["clip"], 128), g["s3_128"], atol=1e-5)
    np.testing.assert_allclose(OL.s3gen_mel_spectrogram(g["y24"]), g["s3gen80"], atol=1e-4)


WHISPER_GOLDEN = dict(dims="micro.en", seed=77, style="peaky", n_new=64, wkeys=["encoder.conv1.weight", "decoder.token_embedding.weight", "decoder.ln.weight"])


def test_oracle_whisper_golden():
    from oracle import whisper as OW
    g = _load("whisper_micro_en_f16.npz")
    dims = OW.DIMS[WHISPER_GOLDEN["dims"]]
    w = OW.synthetic_weights(dims, seed=WHISPER_GOLDEN["seed"], style=WHISPER_GOLDEN["style"], round_to="f16")
    np.testing.assert_allclose(_wsum(w, WHISPER_GOLDEN["wkeys"]), g["wsum"], rtol=1e-12)
    ora = OW.WhisperOracle(dims, w)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    xa = ora.encode(g["mel"])
    np.testing.assert_allclose(xa.numpy()[:, :8], g["features"], atol=2e-4)
    oo = OW.DecodingOptions(timestamps=True, suppress_ids=OW.synthetic_suppress_list(st), blank_ids=[220, 50256 - 1], max_new_tokens=WHISPER_GOLDEN["n_new"])
    for b in range(g["tokens"].shape[0]):
        r = OW.greedy_decode(ora, st, xa[b:b + 1], oo)
        assert r.tokens == g["tokens"][b].tolist()
        assert len(set(r.tokens)) >= 16 and np.isfinite(r.avg_logprob)             # the frozen case is not a degenerate one
        np.testing.assert_allclose(r.avg_logprob, g["avg_logprob"][b], atol=1e-4)
    init, _ = OW.initial_tokens(st, oo)
    lg = OW.teacher_forced_logits(ora, xa[0:1], init + g["tokens"][0].tolist())[g["logit_pos"]][:, ::97]
    np.testing.assert_allclose(lg, g["logits_clip0"], atol=2e-4)


def test_oracle_codecs_golden():
    from oracle import codec as OC
    g = _load("codecs_micro.npz")
    cfg = S.SNAC_CONFIGS["snac_micro"]
    ora = OC.SNACOracle(cfg, S.snac_weights(cfg, seed=3))
    codes = [g[f"snac_codes{i}"].tolist() for i in range(len(cfg.vq_strides))]
    np.testing.assert_allclose(ora.decode(codes, g["snac_noise"]), g["snac_pcm"], atol=1e-5)
    dcfg = S.DAC_CONFIGS["dac_micro"]
    np.testing.assert_allclose(OC.DACOracle(dcfg, S.dac_weights(dcfg, seed=4)).decode_from_codes(g["dac_codes"]), g["dac_pcm"], atol=1e-5)


def test_oracle_cosyvoice2_golden():
    from oracle import flow as OF, hift as OH, s3tok as OS
    g = _load("cosyvoice2_micro.npz")
    scfg = S.S3_CONFIGS["s3_micro"]
    ids, n, _ = OS.S3Oracle(scfg, S.s3_weights(scfg, 2)).quantize(g["s3_mel128"][None], np.asarray([g["s3_mel128"].shape[1]]))
    np.testing.assert_array_equal(ids[0, :n[0]], g["s3_ids"])
    fcfg = S.FLOW_CONFIGS["flow_micro"]
    mel, _ = OF.inference(S.flow_weights(fcfg, seed=0), fcfg, g["flow_token"], g["flow_prompt_token"], g["flow_prompt_feat"], g["flow_embedding"], g["flow_z"])
    np.testing.assert_allclose(mel, g["flow_mel"], atol=2e-4)
    hcfg = S.HIFT_CONFIGS["hift_micro"]
    hw = S.hift_weights(hcfg, seed=0)
    np.testing.assert_allclose(OH.f0_predictor(hw, g["hift_mel"]), g["hift_f0"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(OH.source(hw, hcfg, g["hift_f0"], g["hift_noise"]), g["hift_source"], atol=1e-6)
    np.testing.assert_allclose(OH.decode(hw, hcfg, g["hift_mel"], g["hift_source"]), g["hift_pcm"], atol=5e-5)


def test_oracle_lm_golden():
    from oracle import lm as OLM
    g = _load("lm_llama_micro_f16.npz")
    cfg = S.LM_CONFIGS["llama-micro"]
    w = S.lm_weights(cfg, seed=1, round_to="f16")
    np.testing.assert_allclose(_wsum(w, ["model.embed_tokens.weight"]), g["wsum"], rtol=1e-12)
    logits = OLM.LMOracle(cfg, w).forward(g["ids"].astype(np.int64))
    np.testing.assert_allclose(np.asarray(logits[-1], np.float32), g["last_logits"], atol=2e-4)


def test_oracle_campplus_golden():
    from oracle import campplus as OCP
    g = _load("campplus.npz")
    w = S.campplus_weights(2)
    np.testing.assert_allclose(_wsum(w, ["head.conv1.weight", "blocks.2.layers.15.cam_layer.linear_local.weight", "dense.linear.weight"]), g["wsum"], rtol=1e-12)
    np.testing.assert_allclose(OCP.kaldi_fbank(g["clip"]), g["fbank"], atol=1e-4)
    np.testing.assert_allclose(OCP.CAMPPlusOracle(w).inference(g["clip"])[0], g["embedding"], atol=2e-4 * np.abs(g["embedding"]).max())


# ---- GPU: HIP path vs frozen ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_logmel_golden(ctx):
    from mlx_swift_audio_amd import audio as A
    g = _load("logmel.npz")
    np.testing.assert_allclose(A.whisper_log_mel_spectrogram(ctx, g["clip"], 80), g["whisper80"], atol=1e-3)
    np.testing.assert_allclose(A.whisper_log_mel_spectrogram(ctx, g["clip"], 128), g["whisper128"], atol=1e-3)
    np.testing.assert_allclose(A.s3_log_mel_spectrogram(ctx, g["clip"], 128), g["s3_128"], atol=1e-3)
    np.testing.assert_allclose(A.s3gen_mel_spectrogram(ctx, g["y24"]), g["s3gen80"], atol=2e-3)


@pytest.mark.gpu
def test_hip_whisper_golden(ctx):
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import whisper as HW
    from oracle import whisper as OW          # synthetic checkpoint + special-token arithmetic only; the expected values come from the file
    g = _load("whisper_micro_en_f16.npz")
    dims = OW.DIMS[WHISPER_GOLDEN["dims"]]
    model = HW.WhisperModel.load(ctx, dims, OW.synthetic_weights(dims, seed=WHISPER_GOLDEN["seed"], style=WHISPER_GOLDEN["style"], round_to="f16"), m.F16)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    model.encode(g["mel"])
    err = np.abs(model.audio_features()[:, :8] - g["features"])
    assert err.max() <= 0.01 and err.mean() <= 0.0015
    o = HW.DecodingOptions(timestamps=True, suppress_ids=OW.synthetic_suppress_list(st), blank_ids=[220, 50256 - 1], max_new_tokens=WHISPER_GOLDEN["n_new"])
    model.trace_logits([0])
    res = HW.GreedyDecoder(model, o).decode(g["mel"])
    assert float(g["margins"].min()) >= 0.03                 # every frozen step is resolvable in f16 (logit noise ~1e-3, test_whisper_steps_gpu.py)
    for b in range(g["tokens"].shape[0]):
        assert res[b].tokens == g["tokens"][b].tolist()     # all 64 ids of all 4 clips, no fork rule
        np.testing.assert_allclose(res[b].avg_logprob, g["avg_logprob"][b], atol=3e-3)
        np.testing.assert_allclose(res[b].no_speech_prob, g["no_speech_prob"][b], rtol=0.02, atol=1e-7)
    # the step graph's own logits at the frozen positions (generated index p -> trace row n_init - 1 + p, n_init = 1)
    for i, p in enumerate(g["logit_pos"]):
        row = model.read_logit_trace(0, int(p), 1)[0, ::97]
        assert np.abs(row - g["logits_clip0"][i]).max() <= 0.012 * np.abs(g["logits_clip0"][i]).max()
    model.close()


@pytest.mark.gpu
def test_hip_codecs_golden(ctx):
    from mlx_swift_audio_amd import codec as HC
    g = _load("codecs_micro.npz")
    cfg = S.SNAC_CONFIGS["snac_micro"]
    dec = HC.SNACDecoder.load(ctx, cfg, S.snac_weights(cfg, seed=3))
    codes = [g[f"snac_codes{i}"].tolist() for i in range(len(cfg.vq_strides))]
    np.testing.assert_allclose(dec.decode(codes, g["snac_noise"]), g["snac_pcm"], atol=2e-4)
    dec.close()
    dcfg = S.DAC_CONFIGS["dac_micro"]
    dd = HC.DACCodec.load(ctx, dcfg, S.dac_weights(dcfg, seed=4))
    np.testing.assert_allclose(dd.decode_from_codes(g["dac_codes"][None])[0], g["dac_pcm"], atol=2e-4)
    dd.close()


@pytest.mark.gpu
def test_hip_cosyvoice2_golden(ctx):
    from mlx_swift_audio_amd import flow as HF, hift as HH, s3tok as HS
    g = _load("cosyvoice2_micro.npz")
    scfg = S.S3_CONFIGS["s3_micro"]
    tok = HS.S3Tokenizer.load(ctx, scfg, S.s3_weights(scfg, 2))
    ids, n = tok.quantize(g["s3_mel128"][None], [g["s3_mel128"].shape[1]])
    assert n[0] == g["s3_ids"].shape[0] and (ids[0, :n[0]] == g["s3_ids"]).mean() >= 0.97      # FSQ rounding boundaries, see test_s3tok_gpu.py
    tok.close()
    fcfg = S.FLOW_CONFIGS["flow_micro"]
    fm = HF.FlowModule.load(ctx, fcfg, S.flow_weights(fcfg, seed=0))
    mel = fm.inference(g["flow_token"], g["flow_prompt_token"], g["flow_prompt_feat"], g["flow_embedding"], g["flow_z"])
    np.testing.assert_allclose(mel, g["flow_mel"], atol=2e-3, rtol=2e-3)
    fm.close()
    hcfg = S.HIFT_CONFIGS["hift_micro"]
    hg = HH.HiFTGenerator.load(ctx, hcfg, S.hift_weights(hcfg, seed=0))
    np.testing.assert_allclose(hg.f0_predictor(g["hift_mel"]), g["hift_f0"], rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(hg.m_source(g["hift_f0"], g["hift_noise"]), g["hift_source"], atol=5e-6)
    np.testing.assert_allclose(hg.decode(g["hift_mel"], g["hift_source"]), g["hift_pcm"], atol=3e-4, rtol=1e-3)
    hg.close()


@pytest.mark.gpu
def test_hip_lm_golden(ctx):
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import lm as HL
    g = _load("lm_llama_micro_f16.npz")
    cfg = S.LM_CONFIGS["llama-micro"]
    model = HL.CausalLM.load(ctx, cfg, S.lm_weights(cfg, seed=1, round_to="f16"), m.F16)
    logits = model.forward(g["ids"].tolist())
    np.testing.assert_allclose(np.asarray(logits, np.float32).reshape(-1), g["last_logits"], atol=0.03, rtol=0.02)
    model.close()


@pytest.mark.gpu
def test_hip_campplus_golden(ctx):
    from mlx_swift_audio_amd import speaker as SP
    g = _load("campplus.npz")
    enc = SP.CAMPlusSpeakerEncoder.load(ctx, S.campplus_weights(2))
    np.testing.assert_allclose(enc.extract_fbank(g["clip"]), g["fbank"], atol=2e-3)
    assert np.abs(enc(g["clip"])[0] - g["embedding"]).max() <= 3e-3 * np.abs(g["embedding"]).max()
    enc.close()
