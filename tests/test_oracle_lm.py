"""CPU tests pinning the LM oracle (oracle/lm.py) against the independent `transformers` Llama / Qwen2 implementations
(random init, same weights, same RoPE scaling) and hand-built sampler / parser cases."""
import numpy as np
import pytest
import torch

from mlx_swift_audio_amd import synthetic as S
from oracle import lm as OL


def _hf(cfg, weights):
    sd = {k: torch.from_numpy(v) for k, v in weights.items()}
    if cfg.qkv_bias:
        from transformers import Qwen2Config, Qwen2ForCausalLM
        hc = Qwen2Config(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                         num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                         tie_word_embeddings=True, max_position_embeddings=cfg.max_ctx, attn_implementation="eager")
        m = Qwen2ForCausalLM(hc)
    else:
        from transformers import LlamaConfig, LlamaForCausalLM
        hc = LlamaConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                         num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, head_dim=cfg.head_dim, rms_norm_eps=cfg.rms_eps,
                         rope_theta=cfg.rope_theta, tie_word_embeddings=True, max_position_embeddings=cfg.max_ctx, attn_implementation="eager",
                         rope_scaling={"rope_type": "llama3", "factor": cfg.rope_factor, "low_freq_factor": cfg.rope_low,
                                       "high_freq_factor": cfg.rope_high, "original_max_position_embeddings": cfg.rope_old_ctx})
        m = LlamaForCausalLM(hc)
    sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    return m.eval()


@pytest.mark.parametrize("name", ["llama-micro", "qwen-micro"])
def test_forward_matches_transformers(name):
    cfg = S.LM_CONFIGS[name]
    w = S.lm_weights(cfg, seed=1)
    o = OL.LMOracle(cfg, w)
    ids = [5, 17, 256, 999, 2048, 3, 42]
    lo = o.forward(ids).numpy()
    with torch.no_grad():
        lh = _hf(cfg, w)(torch.tensor([ids])).logits[0].numpy()
    np.testing.assert_allclose(lo, lh, atol=2e-3, rtol=1e-3)
    # incremental == full
    o.reset()
    inc = np.stack([o.forward([t])[0].numpy() for t in ids])
    np.testing.assert_allclose(inc, lo, atol=2e-3, rtol=1e-3)


def test_llama3_freqs_bands():
    f = OL.llama3_freqs(128, 500000.0, True, 32.0, 1.0, 4.0, 8192.0)
    base = OL.llama3_freqs(128, 500000.0, False, 32.0, 1.0, 4.0, 8192.0)
    wl = 2 * np.pi * base
    assert np.allclose(f[wl < 2048], base[wl < 2048])                 # high-frequency band untouched
    assert np.allclose(f[wl > 8192], base[wl > 8192] * 32.0)          # low-frequency band scaled
    mid = (wl > 2048) & (wl < 8192)
    assert mid.any() and np.all(f[mid] > base[mid]) and np.all(f[mid] < base[mid] * 32.0)


def test_top_p_keeps_first_crossing_token():
    logits = np.log(np.array([0.5, 0.3, 0.15, 0.05], np.float32))
    f = OL.top_p_filter(logits, [], 1.0, 1.0, 0.7)
    assert np.isfinite(f[:2]).all() and np.isinf(f[2:]).all()         # 0.5 < 0.7, 0.8 crosses -> keep both, drop the rest
    f = OL.top_p_filter(logits, [], 1.0, 1.0, 0.4)
    assert np.isfinite(f[0]) and np.isinf(f[1:]).all()
    # repetition penalty divides positive / multiplies negative logits of the history
    f = OL.top_p_filter(np.array([2.0, -2.0, 0.5], np.float32), [0, 1, 1], 2.0, 1.0, 1.0)
    np.testing.assert_allclose(f, [1.0, -4.0, 0.5])
    assert OL.sample_with_uniform(np.log(np.array([0.5, 0.3, 0.2])), 0.49) == 0
    assert OL.sample_with_uniform(np.log(np.array([0.5, 0.3, 0.2])), 0.51) == 1
    assert OL.sample_with_uniform(np.array([0.0, -np.inf, 0.0]), 0.75) == 2


def test_parse_output_frames():
    off = OL.CODE_OFFSET
    frame = [off + 7, off + 4096 + 1, off + 2 * 4096 + 2, off + 3 * 4096 + 3, off + 4 * 4096 + 4, off + 5 * 4096 + 5, off + 6 * 4096 + 6]
    toks = [1, 2, OL.AUDIO_CODE_DATA_START_MARKER] + frame + frame[:3] + [OL.END_TOKEN]
    assert OL.parse_output(toks) == [[7], [1, 4], [2, 3, 5, 6]]
