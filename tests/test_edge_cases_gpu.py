"""Smallest legal inputs and empty inputs of every stage (GPU, through the C ABI): one SNAC frame, one DAC step, one flow token with no
prompt, S3 clips of 2-10 mel frames, the shortest log-mel clips; empty inputs must fail loudly with MiaError, never crash."""
import numpy as np
import pytest

from mlx_swift_audio_amd import synthetic as S

pytestmark = pytest.mark.gpu


def test_minimal_sizes(ctx):
    from mlx_swift_audio_amd import audio as A, codec as HC, flow as HF, s3tok as HS
    from oracle import codec as OC, flow as OF, logmel as OL, s3tok as OS
    rng = np.random.default_rng(0)
    cfg = S.SNAC_CONFIGS["snac_micro"]
    w = S.snac_weights(cfg, 3)
    dec, ora = HC.SNACDecoder.load(ctx, cfg, w), OC.SNACOracle(cfg, w)
    codes = [rng.integers(0, cfg.codebook_size, cfg.vq_strides[0] // s).tolist() for s in cfg.vq_strides]       # one frame
    np.testing.assert_allclose(dec.decode(codes), ora.decode(codes), atol=2e-4)
    dec.close()
    dcfg = S.DAC_CONFIGS["dac_micro"]
    dw = S.dac_weights(dcfg, 4)
    dd = HC.DACCodec.load(ctx, dcfg, dw)
    c1 = rng.integers(0, dcfg.codebook_size, (1, dcfg.n_codebooks, 1))                                           # one code step
    np.testing.assert_allclose(dd.decode_from_codes(c1)[0], OC.DACOracle(dcfg, dw).decode_from_codes(c1[0]), atol=2e-4)
    dd.close()
    fcfg = S.FLOW_CONFIGS["flow_micro"]
    fw = S.flow_weights(fcfg)
    fm = HF.FlowModule.load(ctx, fcfg, fw)
    tok = np.asarray([5], np.int32)                                                                               # one token, no prompt
    z = rng.standard_normal((80, 2)).astype(np.float32)
    emb = rng.standard_normal(fcfg.spk_embed_dim).astype(np.float32)
    got = fm.inference(tok, np.zeros(0, np.int32), np.zeros((0, 80), np.float32), emb, z)
    want, _ = OF.inference(fw, fcfg, tok, np.zeros(0, np.int64), np.zeros((0, 80), np.float32), emb, z)
    np.testing.assert_allclose(got, want, atol=2e-3, rtol=2e-3)
    fm.close()
    scfg = S.S3_CONFIGS["s3_micro"]
    sw = S.s3_weights(scfg, 2)
    st, so = HS.S3Tokenizer.load(ctx, scfg, sw), OS.S3Oracle(scfg, sw)
    for n in (400, 640, 1600):                                                                                    # 2, 4, 10 mel frames
        mel = A.s3_log_mel_spectrogram(ctx, OL.synth_clip(1, n), scfg.n_mels)
        ids, cnt = st.quantize(mel[None], [mel.shape[1]])
        rid, rn, _ = so.quantize(mel[None], np.asarray([mel.shape[1]]))
        assert cnt[0] == rn[0] and (ids[0, :cnt[0]] == rid[0, :rn[0]]).all()
    st.close()
    short = np.full(401, 0.1, np.float32)
    np.testing.assert_allclose(A.whisper_log_mel_spectrogram(ctx, short, 80), OL.whisper_log_mel_spectrogram(short, 80), atol=1e-3)


def test_empty_inputs_fail_loudly(ctx):
    import mlx_swift_audio_amd as M
    from mlx_swift_audio_amd import audio as A, codec as HC, flow as HF, hift as HH
    with pytest.raises(M.MiaError):
        A.whisper_log_mel_spectrogram(ctx, np.zeros(0, np.float32), 80)
    with pytest.raises(M.MiaError):
        A.whisper_log_mel_spectrogram(ctx, np.zeros(161, np.float32), 80)      # shorter than the 200-sample reflect pad: rejected, not mis-padded
    with pytest.raises(M.MiaError):
        A.resample_audio(ctx, np.zeros(0, np.float32), 24000, 16000)
    cfg = S.SNAC_CONFIGS["snac_micro"]
    dec = HC.SNACDecoder.load(ctx, cfg, S.snac_weights(cfg, 3))
    with pytest.raises((M.MiaError, ValueError)):
        dec.decode([[], [], []])
    dec.close()
    fcfg = S.FLOW_CONFIGS["flow_micro"]
    fm = HF.FlowModule.load(ctx, fcfg, S.flow_weights(fcfg))
    with pytest.raises(M.MiaError):
        fm.inference(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 80), np.float32), np.zeros(fcfg.spk_embed_dim, np.float32), np.zeros((80, 0), np.float32))
    fm.close()
    hcfg = S.HIFT_CONFIGS["hift_micro"]
    hg = HH.HiFTGenerator.load(ctx, hcfg, S.hift_weights(hcfg))
    with pytest.raises(M.MiaError):
        hg(np.zeros((80, 0), np.float32))
    hg.close()
