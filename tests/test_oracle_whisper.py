"""CPU tests pinning the Whisper oracle (oracle/whisper.py).

The reference has no tensor-level goldens ("parity unpinned").  The restatement is cross-checked
against the independent Whisper implementation shipped in `transformers` (random-init, same weights copied in),
and the decode rules against hand-built cases.
"""
import numpy as np
import pytest
import torch

from oracle import whisper as W


def test_special_token_arithmetic():
    # turbo eot 50257, sot 50258, transcribe 50360, timestamp_begin 50365; tiny.en 50256/50257/50363
    m = W.SpecialTokens.for_vocab(51866)
    assert (m.eot, m.sot, m.transcribe, m.timestamp_begin, m.num_languages) == (50257, 50258, 50360, 50365, 100)
    assert m.timestamp_begin + 1501 == 51866
    e = W.SpecialTokens.for_vocab(51864)
    assert (e.eot, e.sot, e.timestamp_begin, e.is_multilingual) == (50256, 50257, 50363, False)
    assert e.timestamp_begin + 1501 == 51864
    assert e.sot_sequence() == [e.sot]
    assert m.sot_sequence(0, "transcribe") == [m.sot, m.sot + 1, m.transcribe]
    b = W.SpecialTokens.for_vocab(51865)   # 99-language multilingual models
    assert (b.transcribe, b.timestamp_begin) == (50359, 50364)


def _hf_model(dims, weights):
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    cfg = WhisperConfig(vocab_size=dims.n_vocab, num_mel_bins=dims.n_mels, encoder_layers=dims.n_audio_layer,
                        encoder_attention_heads=dims.n_audio_head, decoder_layers=dims.n_text_layer,
                        decoder_attention_heads=dims.n_text_head, d_model=dims.n_audio_state,
                        encoder_ffn_dim=4 * dims.n_audio_state, decoder_ffn_dim=4 * dims.n_audio_state,
                        max_source_positions=dims.n_audio_ctx, max_target_positions=dims.n_text_ctx,
                        activation_function="gelu", attn_implementation="eager")
    hf = WhisperForConditionalGeneration(cfg).eval()
    sd = {}
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    sd["model.encoder.conv1.weight"] = t(weights["encoder.conv1.weight"].transpose(0, 2, 1))
    sd["model.encoder.conv1.bias"] = t(weights["encoder.conv1.bias"])
    sd["model.encoder.conv2.weight"] = t(weights["encoder.conv2.weight"].transpose(0, 2, 1))
    sd["model.encoder.conv2.bias"] = t(weights["encoder.conv2.bias"])
    sd["model.encoder.embed_positions.weight"] = t(W.sinusoids(dims.n_audio_ctx, dims.n_audio_state))
    sd["model.encoder.layer_norm.weight"] = t(weights["encoder.ln_post.weight"])
    sd["model.encoder.layer_norm.bias"] = t(weights["encoder.ln_post.bias"])
    sd["model.decoder.embed_tokens.weight"] = t(weights["decoder.token_embedding.weight"])
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    sd["model.decoder.embed_positions.weight"] = t(weights["decoder.positional_embedding"])
    sd["model.decoder.layer_norm.weight"] = t(weights["decoder.ln.weight"])
    sd["model.decoder.layer_norm.bias"] = t(weights["decoder.ln.bias"])

    def attn(src, dst):
        for a, b in (("query", "q_proj"), ("key", "k_proj"), ("value", "v_proj"), ("out", "out_proj")):
            sd[f"{dst}.{b}.weight"] = t(weights[f"{src}.{a}.weight"])
            if a != "key":
                sd[f"{dst}.{b}.bias"] = t(weights[f"{src}.{a}.bias"])

    for side, n in (("encoder", dims.n_audio_layer), ("decoder", dims.n_text_layer)):
        for l in range(n):
            s, d_ = f"{side}.blocks.{l}", f"model.{side}.layers.{l}"
            attn(f"{s}.attn", f"{d_}.self_attn")
            sd[f"{d_}.self_attn_layer_norm.weight"] = t(weights[f"{s}.attn_ln.weight"])
            sd[f"{d_}.self_attn_layer_norm.bias"] = t(weights[f"{s}.attn_ln.bias"])
            if side == "decoder":
                attn(f"{s}.cross_attn", f"{d_}.encoder_attn")
                sd[f"{d_}.encoder_attn_layer_norm.weight"] = t(weights[f"{s}.cross_attn_ln.weight"])
                sd[f"{d_}.encoder_attn_layer_norm.bias"] = t(weights[f"{s}.cross_attn_ln.bias"])
            sd[f"{d_}.fc1.weight"] = t(weights[f"{s}.mlp1.weight"]); sd[f"{d_}.fc1.bias"] = t(weights[f"{s}.mlp1.bias"])
            sd[f"{d_}.fc2.weight"] = t(weights[f"{s}.mlp2.weight"]); sd[f"{d_}.fc2.bias"] = t(weights[f"{s}.mlp2.bias"])
            sd[f"{d_}.final_layer_norm.weight"] = t(weights[f"{s}.mlp_ln.weight"])
            sd[f"{d_}.final_layer_norm.bias"] = t(weights[f"{s}.mlp_ln.bias"])
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("k_proj.bias" in m for m in missing), missing   # HF has no key bias either
    return hf


@pytest.fixture(scope="module")
def micro():
    dims = W.DIMS["micro.en"]
    weights = W.synthetic_weights(dims, seed=3)
    return dims, weights, W.WhisperOracle(dims, weights)


def test_encoder_and_decoder_match_transformers(micro):
    dims, weights, model = micro
    hf = _hf_model(dims, weights)
    rng = np.random.default_rng(0)
    mel = rng.standard_normal((2, 2 * dims.n_audio_ctx, dims.n_mels)).astype(np.float32)
    xa = model.encode(mel)
    with torch.no_grad():
        hf_xa = hf.model.encoder(torch.from_numpy(mel).transpose(1, 2)).last_hidden_state
    np.testing.assert_allclose(xa.numpy(), hf_xa.numpy(), atol=2e-4, rtol=1e-4)
    toks = [50257, 100, 2000, 31]
    logits, _ = model.decode(toks, xa[:1])
    with torch.no_grad():
        hf_logits = hf(encoder_outputs=(hf_xa[:1],), decoder_input_ids=torch.tensor([toks])).logits
    np.testing.assert_allclose(logits.numpy(), hf_logits.numpy(), atol=5e-4, rtol=1e-4)


def test_kv_cache_equals_full_forward(micro):
    dims, weights, model = micro
    mel = np.random.default_rng(1).standard_normal((1, 2 * dims.n_audio_ctx, dims.n_mels)).astype(np.float32)
    xa = model.encode(mel)
    toks = [50257, 50363, 17, 9000, 50400]
    full, _ = model.decode(toks, xa)
    kv = None
    outs = []
    for i, t in enumerate(toks):
        lg, kv = model.decode([t] if i else toks[:1], xa, kv)
        outs.append(lg[0, -1])
    np.testing.assert_allclose(torch.stack(outs).numpy(), full[0].numpy(), atol=2e-4)


def test_greedy_rules_first_token_is_timestamp_and_pairs(micro):
    dims, weights, model = micro
    st = W.SpecialTokens.for_vocab(dims.n_vocab)
    mel = np.random.default_rng(2).standard_normal((1, 2 * dims.n_audio_ctx, dims.n_mels)).astype(np.float32)
    xa = model.encode(mel)
    o = W.DecodingOptions(timestamps=True, suppress_ids=W.synthetic_suppress_list(st), blank_ids=[220], max_new_tokens=24)
    r = W.greedy_decode(model, st, xa, o)
    assert len(r.tokens) <= 24 and r.initial_tokens == [st.sot]
    assert st.timestamp_begin <= r.tokens[0] <= st.timestamp_begin + 50          # first token: timestamp <= 1.0 s
    ts = [t for t in r.tokens if t >= st.timestamp_begin]
    assert ts == sorted(ts)                                                     # monotonic timestamps
    for s in o.suppress_ids:
        assert s not in r.tokens
    assert st.no_timestamps not in r.tokens
    assert 0.0 <= r.no_speech_prob <= 1.0 and (np.isnan(r.avg_logprob) or r.avg_logprob <= 0.0)
    # timestamps off: no_timestamps is appended to the initial sequence and no rule mask applies
    o2 = W.DecodingOptions(timestamps=False, suppress_ids=o.suppress_ids, blank_ids=[220], max_new_tokens=8)
    r2 = W.greedy_decode(model, st, xa, o2)
    assert r2.initial_tokens == [st.sot, st.no_timestamps]


def test_prompt_and_budget(micro):
    dims, weights, model = micro
    st = W.SpecialTokens.for_vocab(dims.n_vocab)
    xa = model.encode(np.zeros((1, 2 * dims.n_audio_ctx, dims.n_mels), np.float32))
    o = W.DecodingOptions(prompt=[11, 12, 13], max_tokens=10, suppress_ids=[], blank_ids=[])
    r = W.greedy_decode(model, st, xa, o)
    assert r.initial_tokens == [st.sot_prev, 11, 12, 13, st.sot]
    assert len(r.tokens) <= 10 - 5                                               # maxGenerate = maxTokens - initial
